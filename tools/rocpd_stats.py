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
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:78]:78s} {c:7d} {t / 1e6:10.2f} {t / c / 1e3:10.1f} {100 * t / tot:6.2f}%")
print(f"{'TOTAL':78s} {sum(a[0] for a in agg.values()):7d} {tot / 1e6:10.2f}")
