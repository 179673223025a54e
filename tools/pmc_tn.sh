#!/bin/bash
# HBM bytes of every weight-gradient GEMM shape, per launch, under several M-split / XCD policies (env tunables of launch_tn).
#   tools/pmc_tn.sh <outdir>   ->  <outdir>/pmc_tn.txt
out=${1:-gpurun_out/pmc_tn}; mkdir -p $out
export TMPDIR=/tmp
: > $out/pmc_tn.txt
i=0
for pol in "" "OD_TN_XCD_MIN_TILES=8" "OD_TN_XCD_MIN_TILES=8 OD_TN_KK=2" "OD_TN_XCD_MIN_TILES=8 OD_TN_KK=4"; do
  i=$((i+1))
  echo "== policy: ${pol:-default}" >> $out/pmc_tn.txt
  for set in FETCH_SIZE; do
    rm -rf $out/p$i
    export $pol > /dev/null 2>&1
    timeout 300 rocprofv3 --pmc $set -d $out/p$i -o res -- python3 tools/mb_tn_all.py > $out/p$i.log 2>&1
    for v in $pol; do unset ${v%%=*}; done
    python3 tools/rocpd_pmc_dispatch.py $(find $out/p$i -name "*.db" | head -1) gemm_tn >> $out/pmc_tn.txt 2>&1
  done
  grep algorithmic $out/p$i.log >> $out/pmc_tn.txt
  rm -rf $out/p$i
done
cat $out/pmc_tn.txt
