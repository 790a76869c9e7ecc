#!/bin/bash
# Submits one gpurun call; resubmits ONLY when no GPU slot was free (status=transient: nothing ran, nothing was charged).
# usage: scratch/gpurun_retry.sh TIMEOUT 'command'
for attempt in $(seq 1 12); do
    out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
    echo "$out" | tail -n 60
    if echo "$out" | grep -q "status=transient"; then echo "[retry] no slot (attempt $attempt), waiting 60 s"; sleep 60; continue; fi
    exit 0
done
exit 3
