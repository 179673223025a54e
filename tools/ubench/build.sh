#!/bin/bash
# Build the micro-benchmarks in-tree (the binaries travel to the GPU box with the snapshot; they are git-ignored).
cd "$(dirname "$0")"
for f in barrier_cost stream_cost mfma_power mix_power mix2_power; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $f $f.hip 2> /dev/null || { echo "hipcc failed on $f.hip" >&2; exit 1; }
done
echo "built: barrier_cost stream_cost mfma_power mix_power mix2_power"
