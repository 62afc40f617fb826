# Tracker pipe utilisation at the bench's launch shapes (headline 15x15 kernel + the other-window legs): rocprofv3 --pmc passes
# (counters only: no other trace domain; the program itself after "--"), then per-kernel medians.
set -e
R=$GRAFT_REPO_ROOT
# libraries (product and oracle) are built before the first profiler line: build tools must not start under the profiler
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu --no-warp --no-host-api --no-hd-share $@"
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_lk_$n -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_lk_$n.log 2>&1 || echo "pmc $n failed"
}
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run b GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run c GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU
# memory-side view of the same launches (texture addresser / L1): how busy the load path is next to the VALU
run d GRBM_GUI_ACTIVE TA_BUSY_avg TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_lk_ | tee $R/gpurun_out/pmc_lk_summary.txt
