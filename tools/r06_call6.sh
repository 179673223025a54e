#!/bin/bash
out=gpurun_out/r06f; mkdir -p $out
V=gpurun_variants
timeout 1200 python3 tools/ab_sustained.py "" "@OD_NT_W4D_MAX_K=512" "@OD_NT_W4D_MAX_K=4096" "$V/libod_w4dx1.so@OD_NT_W4D_MAX_K=512" --rounds=2 > $out/ab_nt_two_sets.txt 2>&1
OD_NT_W4D_MAX_K=4096 timeout 600 python3 -m pytest tests/test_kernels.py -m gpu -q -k "gemm_nt_large_m" > $out/pytest_w4d.txt 2>&1
timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so --rounds=1 > $out/ab_step_base.txt 2>&1
OD_NT_W4D_MAX_K=512 timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so --rounds=1 > $out/ab_step_w4d512.txt 2>&1
OD_NT_W4D_MAX_K=4096 timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so --rounds=1 > $out/ab_step_w4d4096.txt 2>&1
tail -n 12 $out/*.txt
