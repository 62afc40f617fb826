# runs a timing script once per library variant under gpurun_in/.  usage: variants_run.sh tools/<script>.py [args]
set -e
cd $GRAFT_REPO_ROOT
for f in gpurun_in/libdr3hip_*.so; do
  DR3HIP_LIB=$PWD/$f timeout -k 10 200 python "$@" 2>&1 | tail -1
done
