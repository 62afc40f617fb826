# everything profiles/<round> holds: kernel stats + PMC traffic of the bench (LK / pyramid) and of the warp / paste legs,
# and the per-wave instruction counters of the tracker at the bench's launch shape
set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench.sh > $R/gpurun_out/profile_bench.out 2>&1
bash $R/tools/profile_warp.sh > $R/gpurun_out/profile_warp.out 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmc_lkshape -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp --no-frontend --no-host-api > /dev/null 2>&1
python3 $R/bench.py > $R/gpurun_out/bench_default_run.json 2> $R/gpurun_out/bench_default_run.err
python3 $R/bench.py --workload hd --no-frontend --no-host-api --no-warp > $R/gpurun_out/bench_hd_run.json 2> $R/gpurun_out/bench_hd_run.err
