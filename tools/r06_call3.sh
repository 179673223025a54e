#!/bin/bash
# round 6, GPU call 3: the chain's CU-side cost split into loads / stores / the rest; energy per class; the GPU suite on the nt-load default
out=gpurun_out/r06c; mkdir -p $out
V=gpurun_variants
AB_TIMEOUT=300 timeout 1200 python3 tools/ab_bwd_fused.py "" $V/libod_fbx32.so $V/libod_fbx48.so $V/libod_fbx96.so $V/libod_fbx112.so $V/libod_fbx1.so --rounds=2 > $out/chain_price_split.txt 2>&1
timeout 900 python3 tools/energy_classes.py > $out/energy_classes.txt 2>&1
timeout 900 python3 bench.py --steps 10 --warmup 3 --no-extras > $out/bench_noextras.json 2> $out/bench.err
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
tail -n 25 $out/chain_price_split.txt $out/energy_classes.txt; tail -c 1500 $out/bench_noextras.json; tail -n 5 $out/pytest_gpu.txt
