set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_lk1 -o p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp > $R/gpurun_out/pmc_lk1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/pmc_lk2 -o p2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp > $R/gpurun_out/pmc_lk2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU_MUL_I32 --output-format csv -d $R/gpurun_out/pmc_lk3 -o p3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp > $R/gpurun_out/pmc_lk3.log 2>&1 || true
ls $R/gpurun_out/pmc_lk1 $R/gpurun_out/pmc_lk2
