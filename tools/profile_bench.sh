set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo no-cpu.max; grep -m1 "model name" /proc/cpuinfo; free -g | head -2) > $R/gpurun_out/hostinfo.txt 2>&1
# libraries (product and oracle) are built before the first profiler line: build tools must not start under the profiler
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-frontend --no-host-api > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp --no-frontend --no-host-api > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-warp --no-frontend --no-host-api > $R/gpurun_out/pmc_write.log 2>&1
# configs[4]'s per-GPU share (--workload hd): the same three passes ("HBM-roofline run with rocprof counters")
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hd -o bench -- python3 $R/bench.py --workload hd --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/prof_hd.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_hd_fetch -o fetch -- python3 $R/bench.py --workload hd --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_hd_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_hd_write -o write -- python3 $R/bench.py --workload hd --steps 2 --warmup 1 --no-cpu > $R/gpurun_out/pmc_hd_write.log 2>&1
# every leg of the default line but the warp leg (tools/profile_warp.sh has those) and the CPU baseline: detector, post-LK geometry,
# sparse alignment, the other tracker windows, the real-frame leg -- so that every kernel of the product has a duration row
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_all -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-host-api --no-warp --no-hd-share > $R/gpurun_out/prof_all.log 2>&1
ls -R $R/gpurun_out | head -50
