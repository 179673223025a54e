#!/bin/bash
# round 6, GPU call 2: where the chain's 4 ms go (waits / fabric bytes / the CU's own work), store policies with the wait state, nt loads in the step
out=gpurun_out/r06b; mkdir -p $out
V=gpurun_variants
AB_TIMEOUT=300 timeout 1500 python3 tools/ab_bwd_fused.py "" $V/libod_fbx32.so $V/libod_fbx40.so $V/libod_st1.so $V/libod_st2.so $V/libod_st3.so $V/libod_ld2.so $V/libod_ld18.so $V/libod_ld2st2.so --rounds=2 > $out/chain_price_policy.txt 2>&1
OSU_DREAMER_HIP_LIB=$PWD/$V/libod_ld2.so timeout 600 python3 tools/mb_bwd_fused.py > $out/ld2_agreement.txt 2>&1
OSU_DREAMER_HIP_LIB=$PWD/$V/libod_ld2.so timeout 900 python3 tools/soak_fused.py 300 > $out/ld2_soak.txt 2>&1
timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so $V/libod_ld2.so --rounds=2 > $out/ab_step_ld2.txt 2>&1
tail -n 30 $out/*.txt
