#!/bin/bash
out=gpurun_out/r06h; mkdir -p $out
V=gpurun_variants
for rd in 0 1; do
  for lib in osu_dreamer_amd/libosudreamer_hip.so $V/libod_fwd32.so $V/libod_nqt1.so $V/libod_nqt3.so $V/libod_nqt4.so; do
    OSU_DREAMER_HIP_LIB=$PWD/$lib timeout 300 python3 tools/ab_fwd.py 2>&1 | grep -v amdgpu >> $out/ab_fwd16x_nqt.txt
  done
done
timeout 900 python3 tools/ab_sampler.py osu_dreamer_amd/libosudreamer_hip.so $V/libod_fwd32.so $V/libod_nqt1.so > $out/ab_sampler_fwd16x.txt 2>&1
timeout 900 python3 -m pytest tests/test_kernels.py tests/test_full_size.py tests/test_model_parity.py tests/test_sampler50.py -m gpu -q > $out/pytest_gpu_attn.txt 2>&1
tail -n 14 $out/*.txt | grep -v amdgpu.ids
