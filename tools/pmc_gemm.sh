#!/bin/bash
# PMC passes over od_gemm_nt at one shape (separate rocprofv3 runs per counter set, each under its own timeout: a pass with
# TCP_/TCC_ derived counters aborted and then hung in finalisation for the whole gpurun limit).
#   tools/pmc_gemm.sh <outdir> N K     -> <outdir>/pmc_gemm_N_K.txt      (OSU_DREAMER_HIP_LIB selects the library)
out=${1:-gpurun_out/pmc}; N=$2; K=$3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf $out/p$i
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o res -- python3 tools/mb_gemm_one.py $N $K 3 > $out/p$i.log 2>&1
done
: > $out/pmc_gemm_${N}_${K}.txt
for db in $(find $out -name "*.db"); do python3 tools/rocpd_pmc.py $db gemm_nt >> $out/pmc_gemm_${N}_${K}.txt 2>&1; done
