set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "280000 morton" "35000 random" "35000 morton"; do
  set -- $cfg
  d=gpurun_out/tcc_$1_$2
  timeout -k 5 60 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $d -- python3 benchmarks/subm_layer.py --channels 128 --voxels $1 --iters 5 --order $2 > $d.json 2> $d.err
  python3 - "$d" <<PY
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    ids = set()
    for r in csv.DictReader(open(f)):
        if "spconv_wide" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    n = len(ids)
print(sys.argv[1], n, {k: v / max(n, 1) for k, v in agg.items()})
PY
done
