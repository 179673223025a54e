#!/bin/bash
# Cycle accounting of the attention kernels at the bench shape: tools/pmc_attn.sh <outdir> fwd|bwd [lib.so]
out=$1; which=$2; lib=$3; mkdir -p $out
export TMPDIR=/tmp
[ -n "$lib" ] && export OSU_DREAMER_HIP_LIB=$PWD/$lib
export OD_BWD_2STREAM=0
: > $out/pmc_attn_$which.txt
for set in "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
  rm -rf $out/p
  timeout 300 rocprofv3 --pmc $set -d $out/p -o res -- python3 tools/mb_attn_one.py $which 2 > $out/p.log 2>&1
  python3 tools/rocpd_pmc_dispatch.py $(find $out/p -name "*.db" | head -1) flash 2>&1 | tail -3 >> $out/pmc_attn_$which.txt
done
rm -rf $out/p
cat $out/pmc_attn_$which.txt
