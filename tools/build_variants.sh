# builds gpurun_in/libdr3hip_<tag>.so variants for experiments.  usage: build_variants.sh <file.hip> "tag:flags" ...
set -e
cd "$(dirname "$0")/../3dr_amd/csrc"
make -j8 > /dev/null 2>&1
src=$1; shift
mkdir -p ../../gpurun_in /tmp/lkv
rm -f ../../gpurun_in/libdr3hip_*.so
objs=""
for o in dr3hip_api.o pyr_kernels.o lk_kernels.o warp_kernels.o fast_kernels.o align_kernels.o; do [ "$o" != "${src%.hip}.o" ] && objs="$objs $o"; done
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $flags -c $src -o /tmp/lkv/v_$tag.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_in/libdr3hip_$tag.so $objs /tmp/lkv/v_$tag.o
done
ls ../../gpurun_in
