#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) from a rocprofv3 rocpd
sqlite database (`rocprofv3 --kernel-trace --stats` writes <name>_results.db on ROCm 7.2).
usage: tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"unsigned short", "bf16", n)
    n = n.split("(")[0]
    n = re.sub(r"^void ", "", n)
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += e - s
# od_flash_attn_bwd is three kernels (delta, dK/dV, dQ) of which the last two run CONCURRENTLY on two streams (od_flash_attn_bwd_aux):
# their individual start-to-end durations overlap and do not add up.  Per call, the time that matters is the union of the three
# intervals, reported as its own row (calls grouped by their attn_delta launch).
calls, cur = [], None
for n, s_, e_ in sorted(rows, key=lambda r: r[1]):
    if "attn_delta_kernel" in n:
        cur = [s_, e_]
        calls.append(cur)
    elif cur is not None and ("flash_bwd_dkv_kernel" in n or "flash_bwd_dq" in n):
        cur[1] = max(cur[1], e_)
bwd_union = sum(e_ - s_ for s_, e_ in calls)
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:78]:78s} {c:7d} {t / 1e6:10.2f} {t / c / 1e3:10.1f} {100 * t / tot:6.2f}%")
fused = [t for n, (c, t) in agg.items() if "flash_bwd_fused_kernel" in n or "fb_prep_kernel" in n]
nfused = max([c for n, (c, t) in agg.items() if "flash_bwd_fused_kernel" in n] or [0])
if nfused:
    print(f"{'od_flash_attn_bwd_fused per call: fb_prep_kernel + flash_bwd_fused_kernel (one stream)':78s} {nfused:7d} {sum(fused) / 1e6:10.2f} {sum(fused) / nfused / 1e3:10.1f}   (sum of the two rows)")
if calls:
    print(f"{'od_flash_attn_bwd per call: union of delta + dK/dV || dQ (two streams)':78s} {len(calls):7d} {bwd_union / 1e6:10.2f} {bwd_union / len(calls) / 1e3:10.1f}   (rows above overlap)")
print(f"{'TOTAL':78s} {sum(a[0] for a in agg.values()):7d} {tot / 1e6:10.2f}")
