# Round evidence: rocprofv3 kernel stats of the bench command + PMC passes (one counter group per run, no tracing flags with --pmc),
# summarised per kernel. usage: bash scratch/prof_all.sh <tag> [bench args]   → gpurun_out/prof_<tag>/ ; copy what is to be judged
# into profiles/<tag>/ (kernel_stats.csv, pmc_summary.txt, bench_under_rocprof.json, pmc_traffic.json).
set -e
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-accuracy "$@" > $O/bench_under_rocprof.json 2> $O/stats.err
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
export AQ3D_GRAPH=0
run() { name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-accuracy $BENCH_EXTRA > $O/pmc_$name.json 2> $O/pmc_$name.err || { tail -5 $O/pmc_$name.err; exit 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
python3 scratch/pmc_summary.py $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d > $O/pmc_summary.txt
python3 scratch/pmc_traffic.py $O/pmc_c $O/pmc_d > $O/pmc_traffic.json
rm -rf $O/stats $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
head -14 $O/pmc_summary.txt | cut -c1-300
cat $O/pmc_traffic.json
