#!/usr/bin/env python3
"""Per-DISPATCH counter values (in launch order) from a rocprofv3 rocpd sqlite database.
usage: tools/rocpd_pmc_dispatch.py results.db kernel-substring"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2]
rows = db.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection").fetchall()
d = defaultdict(dict)
for n, c, v, did, du in rows:
    if flt in n:
        d[did][c] = d[did].get(c, 0.0) + v
        d[did]["_us"] = du / 1e3
for did in sorted(d):
    print(f"dispatch {did}: " + "  ".join(f"{k}={v:.1f}" for k, v in sorted(d[did].items())))
