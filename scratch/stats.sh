# usage: scratch/stats.sh <outname> [env assignments...]  — kernel stats of a short bench run → gpurun_out/<outname>.txt
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/_stats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_stats -- python3 bench.py --steps 40 --warmup 7 --no-cpu-baseline --no-accuracy $BENCH_ARGS > gpurun_out/_stats.json 2> gpurun_out/_stats.err
python3 - "$name" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
nfr = max(1.0, sum(float(r["Calls"]) for r in rows if r["Name"].startswith("vox_insert")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ["total kernel us/frame %.1f (%d frames)" % (tot / nfr / 1e3, nfr)]
for r in rows[:45]:
    out.append("%-64s n/fr %5.1f  us/fr %7.1f  avg %7.1f" % (r['Name'][:64], float(r['Calls']) / nfr, float(r['TotalDurationNs']) / nfr / 1e3, float(r['AverageNs']) / 1e3))
open("gpurun_out/%s.txt" % sys.argv[1], "w").write("\n".join(out))
PY
rm -rf gpurun_out/_stats
