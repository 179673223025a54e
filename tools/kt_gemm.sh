#!/bin/bash
# per-kernel in-step durations under a library: kt.sh <label> <lib>
export TMPDIR=/tmp; out=gpurun_out/r05r/$1; mkdir -p $out
export OSU_DREAMER_HIP_LIB=$2
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o res -- python3 bench.py --steps 3 --warmup 1 --no-extras > $out/kt.log 2>&1
python3 tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 2>/dev/null | grep "gemm_nt_w4\|gemm_tn_w4\|TOTAL" | sed "s/^/$1 /"
rm -rf $out/kt
