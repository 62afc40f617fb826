set -e
cd $GRAFT_REPO_ROOT
for r in "-DDR3_PANO_LOG=6" "-DDR3_PANO_LOG=5" "-DDR3_PANO_LOG=4" "-DDR3_PANO_LOG=3"; do
  touch 3dr_amd/csrc/warp_kernels.hip
  make -C 3dr_amd/csrc EXTRA="$r" > /dev/null 2>&1
  python tools/warp_quick.py "[$r]" 2>/dev/null
done
touch 3dr_amd/csrc/warp_kernels.hip; make -C 3dr_amd/csrc > /dev/null 2>&1
