# PMC passes over the single-layer microbenchmark (one rocprofv3 run per counter group; no tracing flags with --pmc)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="benchmarks/subm_layer.py --channels ${CH:-128} --voxels ${VOX:-35000} --iters 5 --order ${ORDER:-random}"
run() { name=$1; shift; timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pl_$name -- python3 $ARGS > gpurun_out/pl_$name.json 2> gpurun_out/pl_$name.err || { tail -5 gpurun_out/pl_$name.err; exit 1; }; }
if [ -z "$SKIP_AB" ]; then
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
fi
if [ -n "$PMC_C" ]; then run c $PMC_C; fi
if [ -n "$PMC_D" ]; then run d $PMC_D; fi
if [ -n "$PMC_E" ]; then run e $PMC_E; fi
python3 scratch/pmc_summary.py gpurun_out/pl_a gpurun_out/pl_b gpurun_out/pl_c gpurun_out/pl_d gpurun_out/pl_e > gpurun_out/pl_summary.txt
grep -i "kernel\|spconv" gpurun_out/pl_summary.txt
