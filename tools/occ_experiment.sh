cd $GRAFT_REPO_ROOT
for pad in 0 12000 18000 38000; do
  DR3_LK_LDS_PAD=$pad DR3HIP_LIB=$PWD/gpurun_in/libdr3hip_occ.so timeout -k 10 200 python tools/lk_time.py 2>&1 | tail -1 | sed "s/^/pad $pad: /"
done
