"""ms per 50-step sampler call (B=4, L=1115) in the three compute modes: python tools/sampler_modes.py [reps]"""
import json, os, sys
import torch
sys.path.insert(0, os.getcwd())
import bench
out = bench.sampler_bench(torch.device("cuda:0"))
print(json.dumps({k: v["ms_per_sample_call"] for k, v in out.items() if isinstance(v, dict)}))
