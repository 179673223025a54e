import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from osu_dreamer_amd import ops, _lib
dev = torch.device("cuda:0")
tr = bench.make_trainer(dev, seed=1)
m = tr.diffusion
eng = m.engine
B, L, E, U = 32, 8192, 6, m.args.u_head_dim
g = torch.Generator(device=dev).manual_seed(0)
xt = torch.randn(B, E, L, device=dev, generator=g)
dfm = torch.randn(B, U, device=dev, generator=g)
m.attach_grads()
W, Gs = eng._uhead_w(), eng._uhead_w(grads=True)
fsum = torch.zeros(B, U, device=dev)
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"{os.path.basename(os.environ.get('OSU_DREAMER_HIP_LIB', 'in-tree'))}: uhead_fwd {t(lambda: ops.uhead_fwd(xt, W, fsum, U)):.0f} us, uhead_bwd {t(lambda: ops.uhead_bwd(xt, W, dfm, Gs, U)):.0f} us", flush=True)
