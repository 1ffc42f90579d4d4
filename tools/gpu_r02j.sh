#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02j; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python tools/dbg_image.py 2>&1 | tail -30 ) > $OUT/dbg_image.txt
( timeout 900 python -m pytest tests/test_mdconv_gpu.py tests/test_linear_q_gpu.py -q 2>&1 | tail -25 ) > $OUT/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/hm4_probe.py kernels > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
cat $OUT/dbg_image.txt; tail -12 $OUT/pytest.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "bevops" in r["Name"]: print(r["Name"][:120], r["Calls"], r["AverageNs"])
PY
