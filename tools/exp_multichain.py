"""Experiment (round 3): the 50-step sampler's 4 difficulties as INDEPENDENT chains.  The batch rows of `DiffusionModel.sample` never
interact (models/diffusion/model.py:117-138), so B = 4 can run as C concurrent chains of B/C rows, each with its own workspace, graph
and stream: kernels of different chains overlap, filling the tail rounds the one-chain launch sequence leaves idle (its ~81 kernels per
evaluation are 5-33 us each with 1.1-1.6 block rounds).  Prints ms per 50-step call for C = 1, 2, 4 in the three compute modes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from osu_dreamer_amd.model import DiffusionModel


def main():
    dev = torch.device("cuda:0")
    a = bench.default_model_args()
    torch.manual_seed(5)
    base = DiffusionModel(a["emb_dim"], a["a_dim"], a["style_dim"], a["diffusion_args"])
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in base.named_parameters():
            if any(z in n for z in ("ssg1.", "ssg2.", "proj_out.", "u_mod.")) or n == "u_out.weight":
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    sd = base.state_dict()
    B, L = 4, 1115
    h = torch.randn(1, 128, L, generator=g).to(dev)
    s = torch.randn(B, 32, generator=g).to(dev)
    x_init = torch.randn(B, 6, L, generator=g).to(dev)
    models = []
    for _ in range(4):
        m = DiffusionModel(a["emb_dim"], a["a_dim"], a["style_dim"], a["diffusion_args"])
        m.load_state_dict(sd)
        models.append(m.to(dev))
    streams = [torch.cuda.Stream(dev) for _ in range(4)]
    for name, dt, mm in (("bf16", torch.bfloat16, "f32"), ("fp32_bf16x3", torch.float32, "bf16x3"), ("fp32", torch.float32, "f32")):
        ref = None
        for C in (1, 2, 4):
            nb = B // C
            for m in models:
                m.compute_dtype, m.f32_matmul = dt, mm

            def run():
                outs = []
                cur = torch.cuda.current_stream(dev)
                for c in range(C):
                    streams[c].wait_stream(cur)
                    with torch.cuda.stream(streams[c]):
                        outs.append(models[c].sample(h, s[c * nb:(c + 1) * nb], 50, x_init=x_init[c * nb:(c + 1) * nb]))
                for c in range(C):
                    cur.wait_stream(streams[c])
                return torch.cat(outs, 0)
            run(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.time(); x = run(); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
            if ref is None:
                ref = x
            err = float((x - ref).norm() / ref.norm())
            print(f"{name:12s} chains={C} rows/chain={nb}: {min(ts):7.2f} ms per 50-step call (runs {' '.join('%.1f' % t for t in ts)}), rel diff vs one chain {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
