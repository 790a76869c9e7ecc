#!/bin/bash
# Runs "name|seconds|command" steps one after the other; an ordinary failure (assertion, non-zero exit) lets the next step run,
# a timeout or a signal (rc 124 / >= 128) stops the whole call — no GPU step is started after a killed one.
mkdir -p gpurun_out
for step in "$@"; do
    name="${step%%|*}"; rest="${step#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
    echo "=== [$name] (limit ${secs}s) $(date +%T)"
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] rc=$rc $(date +%T)"; tail -n 4 "gpurun_out/$name.log"
    if [ "$rc" -eq 124 ] || [ "$rc" -ge 128 ]; then echo "=== stopping: $name was killed (rc=$rc)"; exit "$rc"; fi
done
exit 0
