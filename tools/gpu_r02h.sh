#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest_all.log
( timeout 300 python tools/bevdet_slice.py 2>&1 | grep "{" ) > $OUT/bevdet_slice.jsonl
( timeout 600 python tools/hm4_probe.py 2>&1 | grep "{" | grep -v '"variant": 17[0-9]' ) > $OUT/hm4_probe.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/hm4_probe.py kernels > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -12 $OUT/pytest_all.log; cat $OUT/bevdet_slice.jsonl; grep base_sca $OUT/hm4_probe.jsonl | grep uniform
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    print(r["Name"][:120], r["Calls"], r["AverageNs"])
PY
