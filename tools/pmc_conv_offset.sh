#!/bin/bash
# PMC pass over tools/conv_offset_probe.py (FETCH_SIZE / TCC hit+miss per kernel).  Needs ~3-4 min on a fresh box
# (first torch import + counter serialisation): the one attempt of round 1 ran out of its 110 s budget.
OUT=gpurun_out/r01f_pmc; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p -o k -- python $GRAFT_REPO_ROOT/tools/conv_offset_probe.py 2>&1 | tail -3 ) > $OUT/log.txt
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][-60:]
        agg[(n, r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, gsz), c in sorted(agg.items()):
    if "conv3x3" in n or "dcn_glds" in n:
        print(n, "grid", gsz, {k: round(sum(v) / len(v), 1) for k, v in c.items()}, "calls", len(next(iter(c.values()))))
PY
cat $OUT/log.txt | tail -2
