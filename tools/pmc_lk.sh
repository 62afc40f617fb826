set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_lk_$n -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp --no-frontend --no-host-api > $R/gpurun_out/pmc_lk_$n.log 2>&1 || echo "pmc $n failed"
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run c GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_lk_ | tee $R/gpurun_out/pmc_lk_summary.txt
