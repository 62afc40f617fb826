# counters of the warp legs (tools/warp_quick.py): issue / wait / cache / traffic, separate --pmc passes
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_warpq_$n -o p -- python3 $R/tools/warp_quick.py > $R/gpurun_out/pmc_warpq_$n.log 2>&1 || echo "pmc $n failed"
}
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run b GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
run c FETCH_SIZE TCC_HIT_sum TCC_MISS_sum
run d WRITE_SIZE TCC_REQ_sum TCC_READ_sum
run e TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avg TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_warpq_ | tee $R/gpurun_out/pmc_warpq_summary.txt
