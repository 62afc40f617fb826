set -e
for i in 1 2; do
DR3HIP_LIB=3dr_amd/variants/libdr3hip_leftall.so timeout -k 10 300 python tools/warp_quick.py 2>&1 | tail -1 | grep -o "stitch [0-9.]* ms" | head -1 | sed "s/$/ leftall/"
timeout -k 10 300 python tools/warp_quick.py 2>&1 | tail -1 | grep -o "stitch [0-9.]* ms" | head -1 | sed "s/$/ shipped/"
done
DR3HIP_LIB=3dr_amd/variants/libdr3hip_leftall.so timeout -k 10 300 python -m pytest tests/test_gpu_stitch.py -x -q -m gpu 2>&1 | tail -2
