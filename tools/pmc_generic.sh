# usage: tools/pmc_generic.sh <tag> <python script + args, relative to the repo root>
# runs several rocprofv3 --pmc passes (counters only: no other trace domain) and prints per-kernel medians
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -o p -- python3 $R/$SCRIPT $SARGS > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo "pmc $n failed"
}
SCRIPT=$1; shift; SARGS="$@"
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD
run c FETCH_SIZE TCC_HIT_sum
run d WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
run e TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avg GRBM_GUI_ACTIVE
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_${TAG}_ | tee $R/gpurun_out/pmc_${TAG}_summary.txt
