#!/bin/bash
# round 6, GPU call 1: what the chain costs today + cache policies of its hand-off; the sampler's wide products on the 4-wave kernel; staggered NT epilogues
out=gpurun_out/r06a; mkdir -p $out
V=gpurun_variants
AB_TIMEOUT=300 timeout 1500 python3 tools/ab_bwd_fused.py "" $V/libod_fbx1.so $V/libod_fbx8.so $V/libod_fbx16.so $V/libod_st1.so $V/libod_st2.so $V/libod_st3.so $V/libod_ld2.so $V/libod_ld17.so --rounds=2 > $out/chain_policy.txt 2>&1
AB_MODES="bf16" timeout 900 python3 tools/ab_sampler.py "" "@OD_NT_BIG_MIN_TILES=190" "@OD_NT_BIG_MIN_TILES=100" > $out/sampler_big_tiles.txt 2>&1
timeout 900 python3 tools/ab_sustained.py "" $V/libod_stag1.so $V/libod_stag2.so --shapes=qkv,vg,d_out,d_proj_o,out --rounds=2 > $out/stagger.txt 2>&1
OD_NT_BIG_MIN_TILES=190 timeout 600 python3 -m pytest tests/test_sampler50.py -m gpu -q > $out/sampler50_big_tiles.txt 2>&1
tail -n 30 $out/*.txt
timeout 900 python3 -m pytest tests/test_trajectory.py -m gpu -q -s > gpurun_out/r06a/trajectory_gpu.txt 2>&1; tail -n 12 gpurun_out/r06a/trajectory_gpu.txt
