cd $GRAFT_REPO_ROOT
for l in 6400 6656 6912 7168 7680 8192 8960; do DR3HIP_TILE_LDS=$l timeout -k 10 200 python tools/warp_quick2.py 2>&1 | tail -1 | sed "s/^/lds $l: /"; done
