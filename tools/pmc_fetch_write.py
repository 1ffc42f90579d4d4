#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per kernel (means over the launches) from two rocprofv3 --pmc --output-format csv output
directories (one counter per pass, as the MI355X guide prescribes) as the JSON bench.py's roofline_frame reads its
`traffic` from (KiB as the counters report them; FETCH is doubled by the reader: gfx950 correction).
usage: pmc_fetch_write.py "<source note>" fetch_dir write_dir substring [substring ...] > out.json"""
import collections
import csv
import glob
import json
import re
import sys

note, fetch_dir, write_dir, pats = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
out = {"source": note}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, counter in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            if any(p in n for p in pats):
                per[(n, r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (n, _), v in per.items():
            acc[n][counter].append(v)
for n, cs in acc.items():
    out[n] = {"FETCH_SIZE_KiB_avg": round(sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1), 1),
              "WRITE_SIZE_KiB_avg": round(sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1), 1),
              "launches": len(cs["FETCH_SIZE"])}
print(json.dumps(out, indent=1))
