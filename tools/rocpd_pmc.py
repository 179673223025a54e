#!/usr/bin/env python3
"""Per-kernel PMC counter sums from a rocprofv3 rocpd sqlite database.
usage: tools/rocpd_pmc.py results.db [kernel-substring]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection").fetchall()
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
dur = defaultdict(float)
for n, c, v, d, du in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"unsigned short", "bf16", n).split("(")[0].replace("void ", "")[:60]
    if flt and flt not in n:
        continue
    agg[n][c] += v
    if d not in disp[n]:
        disp[n].add(d)
        dur[n] += du
for n, cs in agg.items():
    if len(n) > 58 and "at::" in n:
        continue
    nd = len(disp[n])
    print(f"{n}  dispatches={nd}  avg_us={dur[n] / nd / 1e3:.1f}")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {v / nd:16.0f}")
