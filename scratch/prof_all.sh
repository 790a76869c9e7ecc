set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof4_bench.json 2> gpurun_out/prof4.err
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc2_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmc2_$name.json 2> gpurun_out/pmc2_$name.err || { tail -5 gpurun_out/pmc2_$name.err; exit 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 scratch/pmc_summary.py gpurun_out/pmc2_a gpurun_out/pmc2_b gpurun_out/pmc2_c gpurun_out/pmc2_d gpurun_out/pmc2_e > gpurun_out/pmc2_summary.txt
head -30 gpurun_out/pmc2_summary.txt | cut -c1-330
