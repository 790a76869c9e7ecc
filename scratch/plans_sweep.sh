cd $GRAFT_REPO_ROOT
: > gpurun_out/plans_sweep.jsonl
for cfg in "4 0" "4 6" "4 8" "3 0" "5 0" "6 0" "6 8" "2 0" "4 0"; do
  set -- $cfg
  AQ3D_STREAMS=$1 python3 bench.py --no-cpu-baseline --no-accuracy --steps 400 --warmup 20 --inflight $2 > /tmp/sw.json 2> /tmp/sw.err || { tail -3 /tmp/sw.err; continue; }
  python3 -c "
import json
d=json.load(open('/tmp/sw.json'))
print(json.dumps(dict(plans=$1, inflight=$2, value=round(d['value'],1), latency_p50=round(d['latency_ms_single_frame']['p50'],3))))" | tee -a gpurun_out/plans_sweep.jsonl
done
