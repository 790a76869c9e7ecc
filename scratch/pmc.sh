set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$name.json 2> gpurun_out/pmc_$name.err || { tail -5 gpurun_out/pmc_$name.err; exit 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
python3 scratch/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d > gpurun_out/pmc_summary.txt
tail -60 gpurun_out/pmc_summary.txt
