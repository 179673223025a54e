#!/bin/bash
# build a variant of the HIP library with extra -D flags for A/B experiments:
#   tools/build_variant.sh name "-DOD_FWD_NQ=4"   ->  gpurun_variants/libod_<name>.so
set -e
cd "$(dirname "$0")/../osu_dreamer_amd/csrc"
name=$1; flags=$2
out=../../gpurun_variants; mkdir -p $out/obj_$name
for s in gemm rowops misc heads optim attn style latent; do
  extra=""; [ "$s" = "attn" ] && extra="-ffinite-math-only $ATTN_FLAGS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c $s.hip -o $out/obj_$name/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $out/obj_$name/*.o -o $out/libod_$name.so
rm -rf $out/obj_$name
echo built $out/libod_$name.so
