#!/bin/bash
# Build libosudreamer_hip.so for gfx950 (in-tree; the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libosudreamer_hip.so
SRCS="gemm.hip rowops.hip misc.hip heads.hip optim.hip attn.hip attn_bwd_fused.hip style.hip latent.hip comm.hip calib.hip det.hip"
OBJS=""
PIDS=""
mkdir -p build
for s in $SRCS; do
  o=build/${s%.hip}.o
  stale=0
  [ -f "$o" ] || stale=1
  for dep in "$s" od_common.h od_tiles.h od_api_internal.h ../../include/osu_dreamer_hip.h build.sh; do
    [ "$dep" -nt "$o" ] && stale=1
  done
  if [ $stale = 1 ]; then
    rm -f "$o"                  # a failed compile must not leave an old object for the link step
    extra=""
    # attn.hip has no NaN/Inf by construction (finite -1e30 mask): lets hipcc drop the canonicalising v_max
    # -fno-slp-vectorize: keeps the softmax row sums as scalar v_add_f32; SLP packs them into v_pk_add_f32, which costs more
    # than the two adds it replaces beside MFMAs (fwd 8.85 -> 8.05 ms in one A/B run, profiles/r02c_ab_attn.txt)
    case "$s" in attn*.hip) extra="-ffinite-math-only -fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$s" -o "$o" ${OD_HIPCC_FLAGS} &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for p in $PIDS; do wait $p || { echo "hipcc failed" >&2; exit 1; }; done
SHA=$(bash ./source_sha.sh)
g++ -O1 -fPIC -DOD_SRC_SHA="\"$SHA\"" -c version.cpp -o build/version.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/version.o -ldl -o $OUT
echo "built $OUT ($SHA)"
