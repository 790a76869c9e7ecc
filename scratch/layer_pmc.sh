# PMC counters of one sparse layer. usage: bash scratch/layer_pmc.sh <tag> [C] [--w16]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lpmc_$TAG
mkdir -p $O
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -- python3 scratch/layer_pmc.py $ARGS > $O/$name.out 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; }
ARGS="$@"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run f TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
python3 scratch/pmc_summary.py $O/a $O/b $O/c $O/d $O/e $O/f | grep -i "kernel \|spconv_c" | cut -c1-600 > $O/summary.txt
rm -rf $O/a $O/b $O/c $O/d $O/e $O/f
cat $O/summary.txt
