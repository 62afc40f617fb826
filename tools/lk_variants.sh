set -e
cd $GRAFT_REPO_ROOT
for v in "" "-DDR3_DBG_NO_DERIV" "-DDR3_DBG_NO_STAGE" "-DDR3_DBG_NO_DERIV -DDR3_DBG_NO_STAGE"; do
  touch 3dr_amd/csrc/lk_kernels.hip
  make -C 3dr_amd/csrc EXTRA="$v" > /dev/null 2>&1
  echo "variant [$v]"
  python tools/lk_sweep.py 2>&1 | grep -E "levels 4 iters  0|levels 4 iters 30|levels 1 iters  0"
done
