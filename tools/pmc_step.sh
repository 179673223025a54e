#!/bin/bash
# Counter passes over ONE training step of the bench workload (separate rocprofv3 runs; counters are never combined with traces):
#   pass 1 FETCH_SIZE   pass 2 WRITE_SIZE   pass 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
#   tools/pmc_step.sh <outdir>  ->  <outdir>/pmc_step.txt, <outdir>/traffic.json
out=${1:-gpurun_out/pmc_step}; mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $out/p$i -o res -- python3 bench.py --steps 1 --warmup 1 --no-extras > $out/p$i.log 2>&1
done
python3 tools/pmc_summary.py $(find $out -name "*.db" | sort) --json $out/traffic.json > $out/pmc_step.txt 2>&1
