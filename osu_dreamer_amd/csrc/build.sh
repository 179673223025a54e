#!/bin/bash
# Build libosudreamer_hip.so for gfx950 (in-tree; the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libosudreamer_hip.so
SRCS="gemm.hip rowops.hip misc.hip heads.hip optim.hip attn.hip style.hip latent.hip"
OBJS=""
mkdir -p build
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ od_common.h -nt "$o" ] || [ od_tiles.h -nt "$o" ] || [ ../../include/osu_dreamer_hip.h -nt "$o" ]; then
    extra=""
    # attn.hip has no NaN/Inf by construction (finite -1e30 mask): lets hipcc drop the canonicalising v_max
    [ "$s" = "attn.hip" ] && extra="-ffinite-math-only"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$s" -o "$o" ${OD_HIPCC_FLAGS} &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $OUT"
