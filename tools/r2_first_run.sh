set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true; grep -m1 "model name" /proc/cpuinfo) > gpurun_out/hostinfo.txt 2>&1
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
timeout -k 10 400 ./tools/ubench/valu_rate gpurun_out/ubench_valu_rate.json > gpurun_out/ubench.log 2>&1
timeout -k 10 300 python3 bench.py > gpurun_out/bench_r2_base.json 2> gpurun_out/bench_r2_base.err
bash tools/pmc_lk.sh > gpurun_out/pmc_lk.out 2>&1
tail -5 gpurun_out/ubench.log; cat gpurun_out/pmc_lk_summary.txt | head -60
