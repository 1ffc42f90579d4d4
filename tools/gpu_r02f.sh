#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/hm4_probe.py kernels > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "msda" in r["Name"]: print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
