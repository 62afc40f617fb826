# usage: tools/pmc_generic.sh <tag> <python script + args, relative to the repo root>
# runs several rocprofv3 --pmc passes (counters only: no other trace domain) and prints per-kernel medians.
# FETCH_SIZE takes 3 of the 4 TCC counter slots and WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots): each has a pass of its
# own.  A pass that fails fails the script.
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
# the libraries are built before the first profiler line: under --pmc every process of a tree is GPU-initialised, and build tools
# must not be started from such a process
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
SCRIPT=$1; shift; SARGS="$@"
run() { n=$1; shift
  if ! timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -o p -- python3 $R/$SCRIPT $SARGS > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1; then
    echo "pmc pass $n ($*) failed: see gpurun_out/pmc_${TAG}_$n.log"; tail -5 $R/gpurun_out/pmc_${TAG}_$n.log; exit 1
  fi
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run f TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avg GRBM_GUI_ACTIVE
run g SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_${TAG}_ | tee $R/gpurun_out/pmc_${TAG}_summary.txt
