#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02m; mkdir -p $OUT; export TMPDIR=/tmp
( BEVOPS_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_mdconv_gpu.py -q -x 2>&1 | tail -30 ) > $OUT/pytest.log
( timeout 300 python tools/dcn_int8_time.py 2>&1 | grep "{" ) > $OUT/dcn_int8_time.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/dcn_int8_time.py > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -15 $OUT/pytest.log; cat $OUT/dcn_int8_time.jsonl
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "dcn" in r["Name"] or "im2col" in r["Name"] or "gemm_tn_s8" in r["Name"] or "conv3x3" in r["Name"] or "nchw" in r["Name"] or "repack" in r["Name"]: print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
