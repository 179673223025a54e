#!/bin/bash
# Build a variant of the HIP library for A/B experiments: only the listed sources are recompiled with the extra flags, the
# rest are the in-tree objects (osu_dreamer_amd/csrc/build/*.o — run csrc/build.sh first).
#   tools/build_variant.sh name "-DOD_FWD32_NQB=2" [attn gemm ...]   ->  gpurun_variants/libod_<name>.so   (default source: attn)
set -e
cd "$(dirname "$0")/../osu_dreamer_amd/csrc"
name=$1; flags=$2; shift 2 || true
srcs=${@:-attn}
out=../../gpurun_variants; mkdir -p $out/obj_$name
objs=""
for s in gemm rowops misc heads optim attn style latent comm vendor_gemm; do
  if [[ " $srcs " == *" $s "* ]]; then
    extra=""; [ "$s" = "attn" ] && extra="-ffinite-math-only ${ATTN_SLP--fno-slp-vectorize}"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $flags -c $s.hip -o $out/obj_$name/$s.o
    objs="$objs $out/obj_$name/$s.o"
  else
    objs="$objs build/$s.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/version.o -ldl -o $out/libod_$name.so
rm -rf $out/obj_$name
echo built $out/libod_$name.so
