set -e
for i in 1 2; do
for v in seg1 seg4; do DR3HIP_LIB=3dr_amd/variants/libdr3hip_$v.so timeout -k 10 300 python tools/warp_quick.py 2>&1 | tail -1 | grep -o "half [0-9.]* ms frac [0-9.]*" | sed "s/$/ $v/"; done
timeout -k 10 300 python tools/warp_quick.py 2>&1 | tail -1 | grep -o "half [0-9.]* ms frac [0-9.]*" | sed "s/$/ seg2/"
done
