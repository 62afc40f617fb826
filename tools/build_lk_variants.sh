# builds gpurun_in/libdr3hip_<tag>.so variants of the tracker for tools/lk_variants.sh.  usage: build_lk_variants.sh "tag:flags" ...
set -e
cd "$(dirname "$0")/../3dr_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../gpurun_in /tmp/lkv
rm -f ../../gpurun_in/libdr3hip_*.so
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $flags -c lk_kernels.hip -o /tmp/lkv/lk_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_in/libdr3hip_$tag.so dr3hip_api.o pyr_kernels.o /tmp/lkv/lk_$tag.o warp_kernels.o fast_kernels.o
done
ls -la ../../gpurun_in
