#!/bin/bash
# All micro-benchmarks behind the round-3 GEMM / attention decisions, one call:  tools/ubench/run_all.sh <outdir>
# (binaries are built in-tree beforehand: tools/ubench/build.sh)
out=${1:-gpurun_out/ubench}; mkdir -p $out
smi() { /opt/rocm/bin/rocm-smi -P -c 2>/dev/null | grep -E "Power|sclk" | sed "s/.*: //" | tr "\n" " "; }
./tools/ubench/barrier_cost > $out/barrier_cost.txt 2>&1
./tools/ubench/stream_cost > $out/stream_cost.txt 2>&1
: > $out/mfma_power.txt
for s in 16 32; do ./tools/ubench/mfma_power $s 4 >> $out/mfma_power.txt & sleep 2.5; echo "[$(smi)]" >> $out/mfma_power.txt; wait; done
: > $out/mix_power.txt
for s in 0 1 2 3 4 5 6 7 8 9 10; do ./tools/ubench/mix_power $s 3.5 >> $out/mix_power.txt & sleep 2.5; echo "[$(smi)]" >> $out/mix_power.txt; wait; done
: > $out/mix2_power.txt
for s in 1 2; do ./tools/ubench/mix2_power $s 3.5 >> $out/mix2_power.txt & sleep 2.5; echo "[$(smi)]" >> $out/mix2_power.txt; wait; done
: > $out/power_probe.txt
for spec in "512 3072 own" "512 3072 vendor" "3072 512 own" "3072 512 vendor" "attn_fwd" "attn_bwd"; do python3 tools/power_probe.py $spec 2>/dev/null | cut -c1-400 >> $out/power_probe.txt; done
OD_NT_W4=0 python3 tools/power_probe.py 512 3072 own 2>/dev/null | cut -c1-400 | sed 's/^/[8-wave kernel] /' >> $out/power_probe.txt
cat $out/*.txt
