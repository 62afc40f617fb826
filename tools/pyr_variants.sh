set -e
cd $GRAFT_REPO_ROOT
for v in "" "-DDR3_DBG_PYR_NOLOAD" "-DDR3_DBG_PYR_NOSTORE" "-DDR3_DBG_PYR_NOLOAD -DDR3_DBG_PYR_NOSTORE"; do
  touch 3dr_amd/csrc/pyr_kernels.hip
  make -C 3dr_amd/csrc EXTRA="$v" > /dev/null 2>&1
  python bench.py --steps 10 --warmup 2 --no-cpu --no-warp 2>/dev/null | python -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant [$v] L0 ms %.4f pass ms %.4f' % (d['roofline']['avg_launch_ms'], d['roofline_pyr_grad_pass']['avg_pass_ms']))"
done
touch 3dr_amd/csrc/pyr_kernels.hip; make -C 3dr_amd/csrc > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_pyr -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-warp > $GRAFT_REPO_ROOT/gpurun_out/pmc_pyr.log 2>&1 || echo pmc failed
