"""Board power and shader clock while one kernel runs back to back for a few seconds:
python3 tools/power_probe.py N K own|vendor   (an NT GEMM at M = 262144; OD_NT_W4=0 selects the 8-wave kernel)   |   attn_fwd | attn_bwd.  Samples `rocm-smi` while the queue drains."""
import os, re, subprocess, sys, time
import torch
sys.path.insert(0, os.getcwd())
from osu_dreamer_amd import ops

dev, bf, M = torch.device("cuda:0"), torch.bfloat16, 32 * 8192
if sys.argv[1] in ("attn_fwd", "attn_bwd"):
    from tools.mb_attn_one import setup
    fwd, bwd, unit = setup(dev)
    which, N, K = sys.argv[1], 0, 0
    run, flops, n = (fwd, 2 * unit, 400) if which == "attn_fwd" else (bwd, 5 * unit, 150)
else:
    N, K, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(M, K, device=dev, generator=g).to(bf)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    C = torch.zeros(M, N, dtype=bf, device=dev)
    run = (lambda: torch.matmul(A, W.t(), out=C)) if which == "vendor" else (lambda: ops.gemm_nt(A, W, None, C))
    flops, n = 2.0 * M * N * K, 4000
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record()
samples = []
t0 = time.time()
while not e1.query() and time.time() - t0 < 20:
    o = subprocess.run(["/opt/rocm/bin/rocm-smi", "-P", "-c"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\):\s*([\d.]+)", o)
    s = re.search(r"sclk clock level:?\s*\d*:?\s*\((\d+)Mhz\)", o)
    samples.append((p.group(1) if p else "?", s.group(1) if s else "?"))
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{which} OD_NT_W4={os.environ.get('OD_NT_W4', '1 (default)')} N={N} K={K}: {ms * 1e3:.0f} us/launch, {flops / ms / 1e9:.0f} TF/s; (W, MHz) samples: {samples[1:-1][:12]}")
if not samples or samples[0][0] == "?":
    print(o[:1500])
