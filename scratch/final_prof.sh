set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err || { tail -5 gpurun_out/final/bench_default.err; exit 1; }
echo default done
python3 bench.py --static --no-cpu-baseline > gpurun_out/final/bench_static.json 2> gpurun_out/final/bench_static.err || { tail -5 gpurun_out/final/bench_static.err; exit 1; }
echo static done
python3 bench.py --workload centerpoint_nusc --no-cpu-baseline > gpurun_out/final/bench_centerpoint_nusc.json 2> gpurun_out/final/bench_nusc.err || { tail -5 gpurun_out/final/bench_nusc.err; exit 1; }
echo nusc done
bash scratch/prof_all.sh final > gpurun_out/final/prof_all.log 2>&1 || { tail -20 gpurun_out/final/prof_all.log; exit 1; }
echo prof done
