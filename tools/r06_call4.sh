#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
V=gpurun_variants
timeout 900 python3 tools/ab_sustained.py "" $V/libod_w4x32.so --shapes=qkv,vg,d_out,d_proj_o --rounds=2 > $out/nt_store_contention.txt 2>&1
timeout 600 python3 -m pytest tests/test_trajectory.py -m gpu -q -s > $out/trajectory_gpu.txt 2>&1
tail -n 12 $out/*.txt
