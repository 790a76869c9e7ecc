# rocprofv3 kernel trace of graph-replayed frames ONLY (bench.py --replay-only): what one frame launches, incl. runtime copy / fill
# kernels. usage: bash scratch/prof_replay.sh <tag> [bench args] -> gpurun_out/prof_<tag>/kernel_stats_replay.csv
set -e
TAG=${1:-r04}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/replay -- python3 bench.py --steps 40 --replay-only "$@" > $O/bench_replay_only.json 2> $O/replay.err
cp $O/replay/*/*kernel_stats.csv $O/kernel_stats_replay.csv
python3 - $O/replay/*/*kernel_trace.csv > $O/frame_sequence.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete frame of one queue: everything between two consecutive ix_mark_points on the same queue
by_q = {}
for r in rows:
    by_q.setdefault(r["Queue_Id"], []).append(r)
last = [r for r in rows if "ix_mark_points" in r["Kernel_Name"]][-1]
q = last["Queue_Id"]
rs = by_q[q]
marks = [i for i, r in enumerate(rs) if "ix_mark_points" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
print("queue", q, "kernels of one replayed frame (between two ix_mark_points):", b - a)
t0 = int(rs[a]["Start_Timestamp"])
for r in rs[a - 3:b]:
    print("%9.1f us  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:110]))
PY
rm -rf $O/replay
cat $O/bench_replay_only.json
