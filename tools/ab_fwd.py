"""Attention forward at the bench shape, sustained (~1.5 s per library): python tools/ab_fwd.py  (library from OSU_DREAMER_HIP_LIB)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.mb_attn_one import setup
from osu_dreamer_amd import _lib
dev = torch.device("cuda:0")
fwd, bwd, unit = setup(dev)
for _ in range(5): fwd()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 150
e0.record()
for _ in range(n): fwd()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{os.path.basename(_lib.loaded_path())}: fwd {ms:.3f} ms {2 * unit / ms / 1e9:.0f} TF/s", flush=True)
