#!/bin/bash
# Build a variant of the HIP library for A/B experiments: only the listed sources are recompiled with the extra flags, the
# rest are the in-tree objects (osu_dreamer_amd/csrc/build/*.o — run csrc/build.sh first).
#   tools/build_variant.sh name "-DOD_FWD32_NQB=2" [attn gemm ...]   ->  gpurun_variants/libod_<name>.so   (default source: attn)
# A source given as  unit=variants/file.hip  compiles that file IN PLACE of unit.hip: the retired kernel variants of earlier rounds
# (csrc/variants/attn_r03_variants.hip: 32x32 dQ, norm + RoPE backward in the epilogues, timing-only modes) build that way, e.g.
#   tools/build_variant.sh dq32 "-DOD_DQ32=1" attn=variants/attn_r03_variants.hip
# (such a library exports that round's C ABI: run it with that round's tools, not through osu_dreamer_amd/_lib.py's header check).
set -e
cd "$(dirname "$0")/../osu_dreamer_amd/csrc"
name=$1; flags=$2; shift 2 || true
srcs=${@:-attn}
out=../../gpurun_variants; mkdir -p $out/obj_$name
objs=""
for s in gemm rowops misc heads optim attn attn_bwd_fused style latent comm calib det; do
  file=""
  for spec in $srcs; do
    [ "$spec" = "$s" ] && file=$s.hip
    case "$spec" in $s=*) file=${spec#*=};; esac
  done
  if [ -n "$file" ]; then
    extra=""; case "$s" in attn*) extra="-ffinite-math-only ${ATTN_SLP--fno-slp-vectorize}";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. $extra $flags -c $file -o $out/obj_$name/$s.o
    objs="$objs $out/obj_$name/$s.o"
  else
    objs="$objs build/$s.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/version.o -ldl -o $out/libod_$name.so
rm -rf $out/obj_$name
echo built $out/libod_$name.so
