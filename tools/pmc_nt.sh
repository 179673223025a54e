#!/bin/bash
# Counters of one long-K NT GEMM shape under the 8-wave kernel, the 4-wave kernel and the vendor library.
#   tools/pmc_nt.sh <outdir> [N K] [sets: "issue" | "mem"]  ->  <outdir>/pmc_nt.txt
out=${1:-gpurun_out/pmc_nt}; N=${2:-512}; K=${3:-3072}; which=${4:-issue}; mkdir -p $out
export TMPDIR=/tmp
: > $out/pmc_nt.txt
if [ $which = issue ]; then
  sets=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS")
else
  sets=("FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE")
fi
for var in own8 own4 vendor; do
  for set in "${sets[@]}"; do
    rm -rf $out/p
    export OD_NT_W4=0; arg=""            # the 4-wave kernel has been the default since round 3: the 8-wave one has to be asked for
    [ $var = own4 ] && export OD_NT_W4=1
    [ $var = vendor ] && arg=vendor
    timeout 300 rocprofv3 --pmc $set -d $out/p -o res -- python3 tools/mb_nt_one.py $N $K $arg > $out/p.log 2>&1
    flt=gemm_nt; [ $var = vendor ] && flt=Cijk
    echo "== $var [$set]" >> $out/pmc_nt.txt
    python3 tools/rocpd_pmc_dispatch.py $(find $out/p -name "*.db" | head -1) $flt 2>&1 | tail -1 >> $out/pmc_nt.txt
  done
done
rm -rf $out/p
cat $out/pmc_nt.txt
