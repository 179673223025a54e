#!/bin/bash
out=gpurun_out/r06i; mkdir -p $out
V=gpurun_variants
for lib in osu_dreamer_amd/libosudreamer_hip.so $V/libod_fwd32.so; do
  OSU_DREAMER_HIP_LIB=$PWD/$lib timeout 600 python3 tools/soak_fwd.py 150 2>&1 | grep -v amdgpu >> $out/soak_fwd.txt
done
for i in 1 2 3; do timeout 300 python3 -m pytest tests/test_model_parity.py -m gpu -q -k "fused_attention_backward_in_the_step or deterministic_mode_matches" 2>&1 | tail -n 4 >> $out/flaky_check_fwd16x.txt; done
for i in 1 2 3; do OSU_DREAMER_HIP_LIB=$PWD/$V/libod_fwd32.so timeout 300 python3 -m pytest tests/test_model_parity.py -m gpu -q -k "fused_attention_backward_in_the_step or deterministic_mode_matches" 2>&1 | tail -n 4 >> $out/flaky_check_fwd32.txt; done
timeout 600 python3 tools/mb_ldc_probe.py 3072 512 2>&1 | grep -v amdgpu > $out/ldc_probe.txt
timeout 600 python3 tools/mb_ldc_probe.py 1024 512 2>&1 | grep -v amdgpu >> $out/ldc_probe.txt
tail -n 30 $out/*.txt
