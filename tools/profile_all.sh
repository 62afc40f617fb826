# everything profiles/<round> holds (tools/summarize_profiles.py condenses it): kernel stats + PMC traffic of the bench (pyramid, LK)
# and of the warp / paste legs, the pipe counters of the tracker kernels, the default and the hd bench lines
set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_bench.sh > $R/gpurun_out/profile_bench.out 2>&1
bash $R/tools/profile_warp.sh > $R/gpurun_out/profile_warp.out 2>&1
bash $R/tools/pmc_lk.sh > $R/gpurun_out/pmc_lk.out 2>&1
cd $R
python3 bench.py > gpurun_out/bench_default_run.json 2> gpurun_out/bench_default_run.err
python3 bench.py --workload hd --no-frontend --no-host-api --no-warp > gpurun_out/bench_hd_run.json 2> gpurun_out/bench_hd_run.err
