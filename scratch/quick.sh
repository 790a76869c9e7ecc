# quick look: bench (no oracle legs) + rocprof kernel stats. usage: bash scratch/quick.sh <tag> [bench args]
set -e
TAG=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/quick_$TAG
mkdir -p $O
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-accuracy "$@" > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-accuracy "$@" > $O/bench_under_rocprof.json 2> $O/stats.err || { tail -20 $O/stats.err; exit 1; }
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
cp $O/stats/*/*kernel_trace.csv $O/kernel_trace.csv
rm -rf $O/stats
python3 - <<P
import json
d = json.load(open("$O/bench.json"))
print("fps", d["value"], "lat", d["latency_ms_single_frame"]["p50"], d["config"].get("row_capacity_per_index_level"))
print({k: round(v, 3) for k, v in d["warmup_breakdown_ms_per_frame"].items()})
P
python3 - <<P
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
frames = None
for r in rows:
    if "ix_count" in r["Name"] or "vox_insert" in r["Name"]:
        frames = int(r["Calls"]); break
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("frames", frames, "kernel ms/frame", tot / 1e6 / frames)
for r in rows[:45]:
    print("%-70s calls/frame %6.2f avg %8.1f us  ms/frame %.3f" % (r["Name"][:70], int(r["Calls"]) / frames, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6 / frames))
P
python3 - <<P
import csv
rows = list(csv.DictReader(open("$O/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full frame on one queue: from the last ix_mark_points back
idx = [i for i, r in enumerate(rows) if "ix_mark_points" in r["Kernel_Name"]]
q = rows[idx[-3]]["Queue_Id"]
seq = [r for r in rows[idx[-3]:] if r["Queue_Id"] == q]
stop = next((j for j, r in enumerate(seq[1:], 1) if "ix_mark_points" in r["Kernel_Name"]), len(seq))
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
for r in seq[:stop]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, r["Kernel_Name"][:60]))
    prev_end = en
P
