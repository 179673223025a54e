"""Joules per training step, by kernel class ("make energy a number").

A class's kernels interleave with the others at millisecond granularity inside a step, finer than any power reading the board offers
(hwmon: ~100 Hz), so the per-class figure is taken the way the sustained GEMM rates are: each class's kernel runs back to back for
~1.5 s at the bench shape (B = 32 x L = 8192, bf16) while bench.PowerSampler reads the board — mean W x ms per launch = J per launch,
x the launches a step makes = J per step of that class.  The sum is printed next to the step's own measured energy (bench.py's
joules_per_step) as the cross-check.

    python3 tools/energy_classes.py [--seconds 1.5]          ->  table + one JSON line
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from osu_dreamer_amd import ops  # noqa: E402

dev, bf = torch.device("cuda:0"), torch.bfloat16
B, L, H, hd, D = 32, 8192, 16, 64, 512
M, dh = B * L, H * hd


def sustained(run, seconds, flops=None):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # size the batch of launches so that the host never runs dry and the region lasts ~seconds
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    n = max(10, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    with bench.PowerSampler(0) as ps:
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    j, src = ps.joules()
    sm = ps.summary()
    return {"ms_per_launch": ms, "watts": (j / (ms * n / 1e3)) if j else sm["power_w_mean"], "clk_mhz": sm["clk_mhz_mean"],
            "j_per_launch": (j / n) if j else None, "tflops": (flops / ms / 1e9) if flops else None, "energy_source": src}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    a = ap.parse_args()
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
    rows = []

    # ---- attention (8 forward + 8 backward launches per step)
    qs, sc = math.log2(math.e) / math.sqrt(hd), 1 / math.sqrt(hd)
    qk, qkv, do = r(M, 2 * dh), r(M, 3 * dh), r(M, dh)
    qk[:, :dh] = (qk[:, :dh].float() * qs).to(bf)
    o = torch.zeros(M, dh, dtype=bf, device=dev)
    lse = torch.zeros(B, H, L, device=dev)
    dqk, dqkv = torch.zeros_like(qk), torch.zeros_like(qkv)
    ws = ops.FusedAttnBwdWorkspace(B, H, L, dev)
    unit = 2.0 * B * H * L * L * hd
    fwd = lambda: ops.flash_attn_fwd(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, lse, B, H, L, hd, sc, q_prescaled=True)
    bwd = lambda: ops.flash_attn_bwd_fused(qk[:, :dh], qk[:, dh:], qkv[:, 2 * dh:], o, do, lse, dqk[:, :dh], dqk[:, dh:], dqkv[:, 2 * dh:],
                                           B, H, L, hd, sc, ws, q_prescaled=True)
    fwd()
    rows.append(("attention backward (od_flash_attn_bwd_fused)", 8, sustained(bwd, a.seconds, 5 * unit)))
    rows.append(("attention forward (od_flash_attn_fwd)", 8, sustained(fwd, a.seconds, 2 * unit)))
    del qk, qkv, do, o, dqk, dqkv, ws
    torch.cuda.empty_cache()

    # ---- NT GEMMs: the eight products of a layer (forward and backward-data), each once per layer
    nt = {"qkv": (3072, 512), "out": (512, 1024), "vg": (2816, 512), "proj_o": (512, 1408), "d_qkv": (512, 3072), "d_vg": (512, 2816),
          "d_out": (1024, 512), "d_proj_o": (1408, 512)}
    acc = {"ms": 0.0, "j": 0.0, "w": 0.0, "fl": 0.0}
    for name, (N, K) in nt.items():
        A_, W_ = r(M, K), (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
        C_ = torch.zeros(M, N, dtype=bf, device=dev)
        s = sustained(lambda: ops.gemm_nt(A_, W_, None, C_), a.seconds / 2, 2.0 * M * N * K)
        acc["ms"] += s["ms_per_launch"]; acc["j"] += (s["j_per_launch"] or 0.0); acc["fl"] += 2.0 * M * N * K
        del A_, W_, C_
    rows.append(("NT GEMMs (8 products of a layer, plain epilogue)", 8,
                 {"ms_per_launch": acc["ms"], "j_per_launch": acc["j"], "watts": acc["j"] / (acc["ms"] / 1e3) if acc["ms"] else None, "clk_mhz": None,
                  "tflops": acc["fl"] / acc["ms"] / 1e9}))

    # ---- TN GEMMs (weight gradients): the four of a layer
    tn = {"w_qkv": (3072, 512), "w_out": (512, 1024), "w_vg": (2816, 512), "w_proj_o": (512, 1408)}
    acc = {"ms": 0.0, "j": 0.0, "fl": 0.0}
    for name, (N, K) in tn.items():
        G_, A_ = r(M, N), r(M, K)
        dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        s = sustained(lambda: ops.gemm_tn(G_, A_, dW, dbias=db), a.seconds / 2, 2.0 * M * N * K)
        acc["ms"] += s["ms_per_launch"]; acc["j"] += (s["j_per_launch"] or 0.0); acc["fl"] += 2.0 * M * N * K
        del G_, A_
    rows.append(("TN GEMMs (4 weight gradients of a layer)", 8,
                 {"ms_per_launch": acc["ms"], "j_per_launch": acc["j"], "watts": acc["j"] / (acc["ms"] / 1e3) if acc["ms"] else None, "clk_mhz": None,
                  "tflops": acc["fl"] / acc["ms"] / 1e9}))

    # ---- row kernels: a streaming pass at the row kernels' own rate (the class is HBM-bound: its energy is its bytes): torch's copy of a
    # [M, 512] bf16 tensor stands in for one read + one write of an activation; the step's row kernels move ~150 GB (profiles/r05_pmc_step.txt)
    x = r(M, D); y = torch.empty_like(x)
    s = sustained(lambda: y.copy_(x), a.seconds)
    gb = 2 * x.numel() * 2 / 1e9
    per_gb_j = (s["j_per_launch"] or 0.0) / gb
    row_gb = 150.0
    rows.append((f"row kernels (streaming at copy speed: {gb / s['ms_per_launch'] * 1e3:.0f} GB/s; {row_gb:.0f} GB per step)", 1,
                 {"ms_per_launch": row_gb / gb * s["ms_per_launch"], "j_per_launch": per_gb_j * row_gb, "watts": s["watts"], "clk_mhz": s["clk_mhz"], "tflops": None}))

    tot_ms = sum(n * v["ms_per_launch"] for _, n, v in rows)
    tot_j = sum(n * (v["j_per_launch"] or 0.0) for _, n, v in rows)
    print(f"{'class':78s} {'launches':>8s} {'ms/step':>8s} {'W':>7s} {'MHz':>6s} {'J/step':>8s} {'share':>6s} {'TF/s':>6s} {'J/PFLOP':>8s}")
    out = []
    for name, n, v in rows:
        j = n * (v["j_per_launch"] or 0.0)
        jpf = (v["j_per_launch"] / (v["tflops"] * v["ms_per_launch"] / 1e6)) if v.get("tflops") and v["j_per_launch"] else None
        print(f"{name:78s} {n:8d} {n * v['ms_per_launch']:8.1f} {v['watts'] or 0:7.0f} {v['clk_mhz'] or 0:6.0f} {j:8.1f} {100 * j / tot_j if tot_j else 0:5.1f}% "
              f"{v['tflops'] or 0:6.0f} {jpf or 0:8.2f}")
        out.append({"class": name, "launches_per_step": n, "ms_per_step": round(n * v["ms_per_launch"], 2), "watts": round(v["watts"] or 0, 1),
                    "joules_per_step": round(j, 1), "joules_per_pflop": round(jpf, 2) if jpf else None})
    print(f"{'sum of classes':78s} {'':8s} {tot_ms:8.1f} {'':7s} {'':6s} {tot_j:8.1f}")
    from osu_dreamer_amd import _lib
    print(json.dumps({"energy_classes": out, "sum_ms_per_step": round(tot_ms, 1), "sum_joules_per_step": round(tot_j, 1),
                      "kernel_src_sha": _lib.source_sha() if hasattr(_lib, "source_sha") else None,
                      "note": "each class sustained ~1.5 s at the bench shape; compare sum_joules_per_step with bench.py's joules_per_step"}))


if __name__ == "__main__":
    main()
