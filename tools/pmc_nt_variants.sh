#!/bin/bash
# Cycle accounting (GRBM cycles, issue stalls, waits) of the 4-wave NT GEMM for a list of library variants.
#   tools/pmc_nt_variants.sh <outdir> N K lib1.so lib2.so ...   (each run with OD_NT_W4=1; "-" = in-tree library, "vendor" = torch.matmul)
out=$1; N=$2; K=$3; shift 3; mkdir -p $out
export TMPDIR=/tmp
: > $out/pmc_variants.txt
for lib in "$@"; do
  for set in "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    rm -rf $out/p
    unset OSU_DREAMER_HIP_LIB; arg=""; flt=gemm_nt
    export OD_NT_W4=1
    if [ "$lib" = vendor ]; then arg=vendor; flt=Cijk; elif [ "$lib" = own8 ]; then export OD_NT_W4=0; elif [ "$lib" != "-" ]; then export OSU_DREAMER_HIP_LIB=$PWD/$lib; fi
    timeout 300 rocprofv3 --pmc $set -d $out/p -o res -- python3 tools/mb_nt_one.py $N $K $arg > $out/p.log 2>&1
    echo "== $lib" >> $out/pmc_variants.txt
    python3 tools/rocpd_pmc_dispatch.py $(find $out/p -name "*.db" | head -1) $flt 2>&1 | tail -1 >> $out/pmc_variants.txt
  done
done
rm -rf $out/p
cat $out/pmc_variants.txt
