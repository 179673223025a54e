#!/bin/bash
# compile attn_bwd_fused.hip for gfx950 and report the scratch (spill) instructions inside its MFMA loops.  usage: tools/fused_spills.sh [extra hipcc flags]
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffinite-math-only -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" -c /root/repo/osu_dreamer_amd/csrc/attn_bwd_fused.hip -o /tmp/abf.o 2>/tmp/abf_res.txt || { grep -i " error" -A4 /tmp/abf_res.txt | head -20; exit 1; }
grep -A14 "Function Name: _ZN12_GLOBAL__N_122flash_bwd_fused_kernelILb1" /tmp/abf_res.txt | grep -i "  VGPRs:\|ScratchSize\|VGPRs Spill\|SGPRs Spill"
awk '/^_ZN12_GLOBAL__N_122flash_bwd_fused_kernelILb1/,/s_endpgm/' attn_bwd_fused-hip-amdgcn-amd-amdhsa-gfx950.s > fused1.s
awk '/^\.LBB/{blk=$1} /scratch_/{c[blk]++} /v_mfma/{m[blk]++} END{for(b in m) printf "%s mfma=%d scratch=%d\n", b, m[b], c[b]}' fused1.s | sort -t= -k2 -n | awk '{printf "%s  ", $0} END{print ""}'
