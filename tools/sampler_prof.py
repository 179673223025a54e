"""Run the configs[3] sampler a few times (for rocprofv3 --kernel-trace):  python3 tools/sampler_prof.py [bf16|f32|x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
from osu_dreamer_amd.model import DiffusionModel
a = bench.default_model_args()
torch.manual_seed(5)
m = DiffusionModel(a["emb_dim"], a["a_dim"], a["style_dim"], a["diffusion_args"])
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for n, p in m.named_parameters():
        if float(p.abs().max()) == 0.0:
            p.copy_(0.02 * torch.randn(p.shape, generator=g))
m = m.to("cuda")
m.use_graph = False
m.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
m.f32_matmul = "bf16x3" if mode == "x3" else "f32"
B, L = 4, 1115
h = torch.randn(1, 128, L, generator=g).cuda()
s = torch.randn(B, 32, generator=g).cuda()
for _ in range(2):
    m.sample(h, s, 50)
torch.cuda.synchronize()
