#!/bin/bash
out=gpurun_out/r06g; mkdir -p $out
V=gpurun_variants
timeout 1500 python3 tools/ab_sustained.py "" $V/libod_stpol1.so $V/libod_stpol2.so $V/libod_stpol3.so $V/libod_stpol4.so $V/libod_plainst.so --shapes=qkv,vg,d_out,d_proj_o --rounds=2 > $out/nt_store_policy.txt 2>&1
for rd in 0 1; do
  for lib in osu_dreamer_amd/libosudreamer_hip.so $V/libod_fwd16x.so; do
    OSU_DREAMER_HIP_LIB=$PWD/$lib timeout 300 python3 tools/ab_fwd.py >> $out/ab_fwd16x.txt 2>&1
  done
done
OSU_DREAMER_HIP_LIB=$PWD/$V/libod_fwd16x.so timeout 900 python3 -m pytest tests/test_kernels.py tests/test_full_size.py -m gpu -q -k "flash_attention or attention_forward or attention_backward_vs" > $out/pytest_fwd16x.txt 2>&1
timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so $V/libod_fwd16x.so --rounds=2 > $out/ab_step_fwd16x.txt 2>&1
tail -n 14 $out/*.txt | grep -v amdgpu.ids
