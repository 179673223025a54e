"""Which forward tensor differs between two identical training forwards (tests/test_model_parity.py::test_fused_attention_backward_in_the_step)?"""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import denoiser_oracle as O
from test_model_parity import make_trainer
dev = torch.device("cuda:0")
d = O.Dims(global_cond_dim=64, backbone_dim=128, n_heads=2, head_dim=64, depth=2, expand=2, radius=1, u_head_dim=16)
P = O.init_params(d, seed=11)
data = O.synthetic_batch(d, 2, 230, seed=12)
snap = []
for run in range(3):
    tr = make_trainer(d, P, dev)
    model = tr.diffusion
    model.compute_dtype = torch.bfloat16
    dd = {k: v.to(dev) for k, v in data.items()}
    opt = tr.configure_optimizers()["optimizer"]
    opt.zero_grad()
    loss, _ = tr(model, dd["h"], dd["z"], dd["s"], None, t=dd["t"], x0=dd["x0"])
    torch.cuda.synchronize()
    ws = model.engine.ws.t
    snap.append({k: v.detach().clone() for k, v in ws.items() if v.dtype in (torch.bfloat16, torch.float32)} | {"loss": loss.detach().clone().reshape(1)})
    print("run", run, "loss", repr(float(loss)))
for run in (1, 2):
    bad = []
    for k in snap[0]:
        a, b = snap[0][k], snap[run][k]
        if a.shape != b.shape: continue
        same = torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a.view(torch.int32), b.view(torch.int16) if b.dtype == torch.bfloat16 else b.view(torch.int32))
        if not same:
            df = (a.float() - b.float()).abs()
            bad.append((k, int((df > 0).sum()), float(df.max())))
    print("run 0 vs run", run, "differing buffers:", bad)
