#!/bin/bash
# round 6: the round's profiles/ records in one GPU call (kernel trace, counters, sampler traces, parity record, bench line), plus energy per class,
# the box's streaming rates, and the GPU test-suite
out=gpurun_out/r06; mkdir -p $out
bash tools/profile_round.sh r06 r06 > $out/profile_round.log 2>&1
timeout 900 python3 tools/energy_classes.py > $out/energy_classes.txt 2>&1
timeout 300 python3 tools/hbm_probe.py > $out/hbm_probe.txt 2>&1
timeout 1800 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -n 4 $out/pytest_gpu.txt; tail -n 8 $out/hbm_probe.txt; tail -c 1200 $out/bench.json
