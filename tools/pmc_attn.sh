#!/bin/bash
# PMC passes over the attention kernels (separate rocprofv3 runs: counters and traces are never combined).
#   tools/pmc_attn.sh <outdir>      -> <outdir>/pmc_attn.txt
out=${1:-gpurun_out/pmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o res -- python3 tools/mb_attn_one.py 2 > $out/p$i.log 2>&1
done
: > $out/pmc_attn.txt
for db in $(find $out -name "*.db"); do python3 tools/rocpd_pmc.py $db flash >> $out/pmc_attn.txt 2>&1; done
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $out/sq_counters.txt
