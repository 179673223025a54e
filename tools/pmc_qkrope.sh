#!/bin/bash
# Counters + timing of the qkv projection's norm + RoPE epilogue kernel (gemm_nt_w4_kernel<2>) under the variant libraries in gpurun_variants/
# (built by tools/build_variant.sh <name> "-DOD_W4Q_X=..." gemm) against the in-tree build and the plain GEMM of the same shape.
#   tools/pmc_qkrope.sh <outdir> [variant names...]   ->  <outdir>/pmc_qkrope.txt
out=${1:-gpurun_out/pmc_qkrope}; shift; mkdir -p $out
export TMPDIR=/tmp
res=$out/pmc_qkrope.txt; : > $res
sets=("FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum")
run() {   # label libpath extra-arg
  echo "==== $1" >> $res
  export OSU_DREAMER_HIP_LIB=$2
  timeout 120 python3 tools/mb_qkrope_one.py $3 time 2>&1 | tail -1 >> $res
  for set in "${sets[@]}"; do
    rm -rf $out/p
    timeout 200 rocprofv3 --pmc $set -d $out/p -o res -- python3 tools/mb_qkrope_one.py $3 > $out/p.log 2>&1
    python3 tools/rocpd_pmc_dispatch.py $(find $out/p -name "*.db" | head -1) gemm_nt_w4 2>&1 | tail -1 >> $res
  done
}
base=$PWD/osu_dreamer_amd/libosudreamer_hip.so
run "plain <0> (in-tree)" $base plain
run "qkrope <2> (in-tree)" $base ""
for v in "$@"; do run "qkrope variant $v" $PWD/gpurun_variants/libod_$v.so ""; done
rm -rf $out/p
cat $res
