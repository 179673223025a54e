#!/bin/bash
# FETCH_SIZE + time of the weight-gradient shapes under the libraries given (variant names under gpurun_variants/, "base" = in-tree)
out=$1; shift; mkdir -p $out; export TMPDIR=/tmp; : > $out/pmc_tn_skew.txt
for v in "$@"; do
  lib=$PWD/osu_dreamer_amd/libosudreamer_hip.so; [ $v != base ] && lib=$PWD/gpurun_variants/libod_$v.so
  export OSU_DREAMER_HIP_LIB=$lib
  echo "== $v" >> $out/pmc_tn_skew.txt
  timeout 200 python3 tools/ab_tn_w4.py 2>/dev/null >> $out/pmc_tn_skew.txt
  rm -rf $out/p; timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/p -o res -- python3 tools/mb_tn_all.py > $out/p.log 2>&1
  python3 tools/rocpd_pmc_dispatch.py $(find $out/p -name "*.db" | head -1) gemm_tn >> $out/pmc_tn_skew.txt 2>&1
done
rm -rf $out/p; cat $out/pmc_tn_skew.txt
