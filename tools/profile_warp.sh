# rocprofv3 kernel stats + PMC traffic of the warp / paste legs (tools/warp_quick.py); separate --pmc passes
set -e
R=$GRAFT_REPO_ROOT
# libraries (product and oracle) are built before the first profiler line: build tools must not start under the profiler
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_warp -o warp -- python3 $R/tools/warp_quick.py > $R/gpurun_out/prof_warp.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_warp_fetch -o fetch -- python3 $R/tools/warp_quick.py > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_warp_write -o write -- python3 $R/tools/warp_quick.py > /dev/null 2>&1
