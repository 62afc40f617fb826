# times every library variant under gpurun_in/ (built in the authoring container with different -D flags) on one box
set -e
cd $GRAFT_REPO_ROOT
for f in gpurun_in/libdr3hip_*.so; do
  DR3HIP_LIB=$PWD/$f timeout -k 10 200 python tools/lk_time.py "$@" 2>&1 | tail -1
done
