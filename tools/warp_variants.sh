set -e
cd $GRAFT_REPO_ROOT
for r in "-DDR3_WARP_NO_XCD_MAP" ""; do
  touch 3dr_amd/csrc/warp_kernels.hip
  make -C 3dr_amd/csrc EXTRA="$r" > /dev/null 2>&1
  python tools/warp_quick.py "[$r]" 2>/dev/null
done
python -m pytest tests/test_gpu_warp.py -m gpu -x -q 2>&1 | tail -2
