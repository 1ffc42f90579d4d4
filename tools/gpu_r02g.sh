#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_msda_hm4_gpu.py tests/test_msda_int8_gpu.py -q 2>&1 | tail -12 ) > $OUT/pytest_hm4.log
( MASKS=0,32,64,96,128,160,224 timeout 600 python tools/hm4_probe.py ablate 2>&1 | grep "{" ) > $OUT/hm4_variants.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/hm4_probe.py kernels > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -5 $OUT/pytest_hm4.log; cat $OUT/hm4_variants.jsonl
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "msda" in r["Name"]: print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
