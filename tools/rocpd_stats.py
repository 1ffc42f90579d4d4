#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 rocpd database (*_results.db): count, average / min / max duration in us, in
dispatch order of first appearance.  usage: rocpd_stats.py file.db [substring ...]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c.lower()][0]
agg = collections.OrderedDict()
for name, start, end in db.execute(f"select {name_col}, start, end from kernels order by start"):
    n = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if pats and not any(p in n for p in pats):
        continue
    agg.setdefault(n, []).append((end - start) / 1e3)
for n, v in agg.items():
    print(f"{n[:90]:90s} x{len(v):<4d} avg {sum(v)/len(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f} us")
