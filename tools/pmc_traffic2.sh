# FETCH_SIZE / WRITE_SIZE of the configs[2] 8UC3 warp, old (direct gather) and strip kernel: separate passes
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export WQ_KITTI=0
for v in new old; do
  if [ $v = old ]; then export DR3HIP_LIB=$R/gpurun_in/libdr3hip_old.so; else unset DR3HIP_LIB; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_t2_${v}_$c -o p -- python3 $R/tools/warp_quick2.py > $R/gpurun_out/pmc_t2_${v}_$c.log 2>&1
  done
  echo "== $v"; python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_t2_${v}_
done
