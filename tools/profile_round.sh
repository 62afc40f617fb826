# One gpurun call that leaves a complete profiles/<round> under gpurun_out/: the profiler passes, the summaries, the tracker's opcode-mix
# files weighted by the counts just measured, and then the bench lines once more WITH those summaries in place (so that they quote the
# traffic and the mix of the kernels they ran).  usage (from the repo root, on the GPU box): bash tools/profile_round.sh r03_v1
# afterwards, in the work tree: cp gpurun_out/profiles_<round>/* profiles/<round>/
set -e
R=$GRAFT_REPO_ROOT
ROUND=${1:-r03_v1}
bash $R/tools/profile_all.sh > $R/gpurun_out/profile_round.out 2>&1
cd $R
mkdir -p profiles/$ROUND
python3 tools/summarize_profiles.py gpurun_out profiles/$ROUND > gpurun_out/profile_round_sum1.log 2>&1
python3 tools/refresh_mix.py profiles/$ROUND
python3 tools/summarize_profiles.py gpurun_out profiles/$ROUND > gpurun_out/profile_round_sum2.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_run.json 2> gpurun_out/bench_default_run.err
python3 bench.py --workload hd --no-frontend --no-host-api --no-warp > gpurun_out/bench_hd_run.json 2> gpurun_out/bench_hd_run.err
DR3_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 32 --no-frontend --no-host-api --no-cpu --no-warp --sharded-paste 2000x1500 > gpurun_out/bench_gpus2_gloo_rehearsal.json 2> gpurun_out/bench_gloo2.err
python3 tools/summarize_profiles.py gpurun_out profiles/$ROUND > gpurun_out/profile_round_sum3.log 2>&1
rm -rf gpurun_out/profiles_$ROUND && cp -r profiles/$ROUND gpurun_out/profiles_$ROUND
python3 -m pytest tests/test_bench_contract.py -q 2>&1 | tail -2
python3 -c "
import json; d=json.loads(open('gpurun_out/profiles_$ROUND/bench_default_run.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['traffic'], 'lk', d['lk_kernel']['avg_launch_ms'], d['lk_kernel']['valu_issue']['frac'])"
