#!/bin/bash
out=gpurun_out/r06e; mkdir -p $out
V=gpurun_variants
AB_TIMEOUT=300 timeout 900 python3 tools/ab_bwd_fused.py "" $V/libod_pack1.so --rounds=3 > $out/chain_pack.txt 2>&1
OSU_DREAMER_HIP_LIB=$PWD/$V/libod_pack1.so timeout 600 python3 tools/mb_bwd_fused.py > $out/pack1_agreement.txt 2>&1
timeout 600 python3 tools/mb_bwd_fused.py > $out/default_agreement.txt 2>&1
OSU_DREAMER_HIP_LIB=$PWD/$V/libod_pack1.so timeout 900 python3 tools/soak_fused.py 200 > $out/pack1_soak.txt 2>&1
timeout 1500 python3 tools/ab_step.py osu_dreamer_amd/libosudreamer_hip.so $V/libod_pack1.so --rounds=2 > $out/ab_step_pack1.txt 2>&1
tail -n 12 $out/*.txt
