cd $GRAFT_REPO_ROOT
export DR3HIP_TILE_LDS=7168
bash tools/variants_run.sh tools/warp_quick2.py
