"""The bench line's `sampler` block alone (configs[3]: 50 steps, B=4, L=1115; fp32 / fp32-bf16x3 / bf16):  python tools/sampler_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.sampler_bench(torch.device("cuda:0")), indent=1))
