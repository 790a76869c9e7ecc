# SQ counters over the whole bench (few steps, eager launches), summarised per kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AQ3D_GRAPH=0
run() { name=$1; shift; timeout -k 5 ${PASS_TIMEOUT:-120} rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pb_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pb_$name.json 2> gpurun_out/pb_$name.err || { tail -5 gpurun_out/pb_$name.err; exit 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
python3 scratch/pmc_summary.py gpurun_out/pb_a gpurun_out/pb_b > gpurun_out/pb_summary.txt
cut -c1-330 gpurun_out/pb_summary.txt | head -24
