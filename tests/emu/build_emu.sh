#!/bin/bash
# Host build of the SAME kernel sources against the SIMT emulator (test infrastructure).
set -e
cd "$(dirname "$0")"
CS=../../osu_dreamer_amd/csrc
OUT=libod_emu.so
mkdir -p build
OBJS=""
PIDS=""
for s in gemm rowops misc heads optim attn attn_bwd_fused style latent comm calib det; do
  o=build/$s.o
  if [ ! -f "$o" ] || [ "$CS/$s.hip" -nt "$o" ] || [ "$CS/od_common.h" -nt "$o" ] || [ "$CS/od_tiles.h" -nt "$o" ] || [ "$CS/od_api_internal.h" -nt "$o" ] || [ ../../include/osu_dreamer_hip.h -nt "$o" ] || [ emu_hip.h -nt "$o" ]; then
    rm -f "$o"
    /opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fPIC -DOD_EMU -DOD_GEMM_BIG_MIN_M=256 -DOD_DW_SMALL_THREADS=10 -DOD_GEMM_SMALL_TILES=6 -DOD_FWD16X_MIN_L=128 -I. -I$CS -Wno-unused-value -c $CS/$s.hip -o $o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for p in $PIDS; do wait $p || { echo "emulator build failed" >&2; exit 1; }; done
SHA=$(bash $CS/source_sha.sh)
# relink only when something changed, and atomically (tmp + mv): parallel test workers may be loading the library meanwhile
stale=0
[ -f "$OUT" ] || stale=1
for o in $OBJS; do [ "$o" -nt "$OUT" ] && stale=1; done
[ -f build/sha.txt ] && [ "$(cat build/sha.txt)" = "$SHA" ] || stale=1
if [ $stale = 1 ]; then
  g++ -O1 -fPIC -DOD_SRC_SHA="\"$SHA\"" -c $CS/version.cpp -o build/version.o
  /opt/rocm/lib/llvm/bin/clang++ -shared -fPIC $OBJS build/version.o -o $OUT.tmp.$$
  mv -f $OUT.tmp.$$ $OUT
  echo "$SHA" > build/sha.txt
fi
echo "built $OUT"
