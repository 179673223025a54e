#!/usr/bin/env python3
"""Per-kernel HBM traffic and MFMA utilisation of a training step from rocprofv3 --pmc passes (rocpd sqlite databases).
usage: tools/pmc_summary.py pass1.db pass2.db pass3.db [--json traffic.json]

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB by rocprofv3; FETCH_SIZE counts 128-byte fabric requests as 64 B on
gfx950, /opt/skills/guides/MI355X_MICROARCH.md section HBM).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: clock = GUI_ACTIVE / 8 / duration,
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 x 1024 SIMDs).
Only the LAST dispatch group of the run (the timed step, after the warm-up step) is averaged: the last `calls/2` dispatches."""
import json, re, sqlite3, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
dbs = [a for a in args if a != jpath]
val = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> [values per dispatch, in order]
dur = defaultdict(list)
for path in dbs:
    db = sqlite3.connect(path)
    seen = set()
    for n, c, v, d, du in db.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection order by dispatch_id"):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"unsigned short", "bf16", n).split("(")[0].replace("void ", "")
        if n.startswith("at::") or "rocprim" in n or "rocclr" in n:
            continue
        val[n][c].append(v)
        if (path, d) not in seen and c in ("FETCH_SIZE", "SQ_BUSY_CYCLES"):
            seen.add((path, d)); dur[n].append(du)

def tail_mean(xs):
    xs = xs[len(xs) // 2:]        # second half = the timed step
    return sum(xs) / max(len(xs), 1)

rows = []
for n, cs in val.items():
    calls = len(cs.get("FETCH_SIZE", [])) // 2
    us = tail_mean(dur[n]) / 1e3 if dur[n] else 0.0
    rd = 2 * tail_mean(cs.get("FETCH_SIZE", [0])) * 1024
    wr = tail_mean(cs.get("WRITE_SIZE", [0])) * 1024
    mf = tail_mean(cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    ga = tail_mean(cs.get("GRBM_GUI_ACTIVE", [0])) / 8.0
    rows.append((n, calls, us, rd, wr, mf / (ga * 1024) if ga else 0.0, ga / (us * 1e3) if us else 0.0))
rows.sort(key=lambda r: -r[1] * r[2])
print(f"{'kernel (per dispatch, timed step)':64s} {'calls':>5s} {'avg_us':>9s} {'read_MB':>9s} {'write_MB':>9s} {'HBM_GB/s':>9s} {'of_8TB/s':>8s} {'MFMA_busy':>9s} {'clk_GHz':>7s}")
for n, calls, us, rd, wr, mfu, clk in rows:
    gbs = (rd + wr) / (us * 1e-6) / 1e9 if us else 0
    print(f"{n[:64]:64s} {calls:5d} {us:9.1f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {gbs:9.0f} {gbs / 8000:8.3f} {mfu:9.3f} {clk:7.2f}")
if jpath:
    byname = {r[0]: r for r in rows}
    def pick(sub):
        return [r for n, r in byname.items() if sub in n]
    bwd = pick("flash_bwd") + pick("attn_delta") + pick("fb_prep")          # od_flash_attn_bwd_fused = fb_prep_kernel + flash_bwd_fused_kernel
    fwd = pick("flash_fwd")
    out = {"B": 32, "L": 8192, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --steps 1 --warmup 1 --no-extras (tools/pmc_step.sh); read = 2 x FETCH_SIZE",
           "od_flash_attn_bwd": {"read_bytes": sum(r[3] for r in bwd), "write_bytes": sum(r[4] for r in bwd), "kernels": [r[0] for r in bwd]},
           "od_flash_attn_fwd": {"read_bytes": sum(r[3] for r in fwd), "write_bytes": sum(r[4] for r in fwd), "kernels": [r[0] for r in fwd]}}
    # kernel classes for the bench line (roofline.also.pmc_classes): time-weighted HBM fraction of the memory-bound kernels,
    # MFMA-pipe busy fraction of the matrix kernels
    def klass(n):
        if "flash_fwd" in n: return "attention_fwd"
        if "flash_bwd" in n: return "attention_bwd"
        if "gemm_nt" in n: return "gemm_nt"
        if "gemm_tn" in n: return "gemm_tn"
        if any(k in n for k in ("rmsnorm", "swiglu", "qk_norm_rope", "dwconv", "attn_delta", "fb_prep", "final_proj", "silu", "adamw", "sqnorm", "cl_to_frames")): return "row_kernels"
        return "other"
    agg = defaultdict(lambda: [0.0, 0.0, 0.0])      # time_us, bytes, mfma-busy x time
    for n, calls, us, rd, wr, mfu, clk in rows:
        a = agg[klass(n)]
        a[0] += calls * us; a[1] += calls * (rd + wr); a[2] += calls * us * mfu
    out["classes"] = {k: {"ms_per_step": round(v[0] / 1e3, 2), "hbm_GBps": round(v[1] / (v[0] * 1e-6) / 1e9, 0) if v[0] else 0,
                          "hbm_frac_of_8TBps": round(v[1] / (v[0] * 1e-6) / 8e12, 3) if v[0] else 0,
                          "mfma_busy": round(v[2] / v[0], 3) if v[0] else 0} for k, v in agg.items()}
    try:      # which build these counters belong to: bench.py quotes the record only while the library it runs has the same hash
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from osu_dreamer_amd import _lib
        out["kernel_src_sha"] = _lib.source_sha()
    except Exception as e:
        out["kernel_src_sha"] = None
        print("could not read the library's source hash:", e)
    json.dump(out, open(jpath, "w"), indent=1)
