"""50-step sampler call (configs[3] shape, hipGraph) per library and mode, one process per library: python tools/ab_sampler.py lib.so ...  (modes: bf16 x3 f32)"""
import os, subprocess, sys, time

def child():
    sys.path.insert(0, os.getcwd())
    import torch, bench
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
    B, L = 4, 1115
    h = torch.randn(1, 128, L, generator=g).cuda()
    s = torch.randn(B, 32, generator=g).cuda()
    out = []
    for mode in os.environ.get("AB_MODES", "bf16 x3 f32").split():
        m.compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
        m.f32_matmul = "bf16x3" if mode == "x3" else "f32"
        for _ in range(2):
            m.sample(h, s, 50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.sample(h, s, 50)
        torch.cuda.synchronize()
        out.append(f"{mode} {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
    print(" | ".join(out))

if os.environ.get("AB_CHILD") == "1":
    child()
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ab_common import parse          # spec = lib.so | lib.so@ENV=VAL,... | @ENV=VAL (in-tree library)
    for rnd in range(2):
        for spec in sys.argv[1:]:
            label, libpath, extra = parse(spec)
            env = dict(os.environ, AB_CHILD="1", OSU_DREAMER_HIP_LIB=libpath, **extra)
            r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=300)
            print(f"[round {rnd}] {label:40s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
