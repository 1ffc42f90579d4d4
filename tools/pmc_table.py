#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 --pmc --output-format csv output directories.
usage: pmc_table.py substring dir [dir ...]   (prints one block per kernel whose name contains the substring)"""
import collections
import csv
import glob
import re
import sys

pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            if pat in n:
                per[(n, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (n, _), cs in per.items():
            for c, v in cs.items():
                acc[n][c].append(v)
for n, cs in acc.items():
    print(n[:100])
    for c in sorted(cs):
        v = cs[c]
        print(f"   {c:36s} {sum(v)/len(v):.4g}  (n={len(v)})")
