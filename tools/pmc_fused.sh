#!/bin/bash
# HBM traffic and L2 hit rate of od_flash_attn_bwd_fused at the bench shape (separate --pmc passes): tools/pmc_fused.sh <outdir> [lib.so]
out=$1; lib=$2; mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export OSU_DREAMER_HIP_LIB=$PWD/$lib
: > $out/pmc_fused.txt
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  rm -rf $out/p
  timeout 300 rocprofv3 --pmc $set -d $out/p -o res -- python3 tools/mb_fused_one.py 2 > $out/p.log 2>&1
  python3 tools/rocpd_pmc.py $(find $out/p -name "*.db" | head -1) flash_bwd_fused 2>&1 | tail -5 >> $out/pmc_fused.txt
done
rm -rf $out/p
cat $out/pmc_fused.txt
