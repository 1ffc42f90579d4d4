#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02q; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mdconv_gpu.py tests/test_model_gpu.py tests/test_ref_kernels_gpu.py tests/test_ref_kernels_live_gpu.py -q -x -k "mdconv or dcn or model or conv" 2>&1 | tail -5 ) > $OUT/pytest.log
( timeout 300 python tools/dcn_int8_time.py 2>&1 | grep "{" ) > $OUT/dcn_int8_time.jsonl
( timeout 300 python tools/ops_timing.py 2>&1 | grep "dcn" ) > $OUT/ops_timing.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/dcn_int8_time.py > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -3 $OUT/pytest.log; grep dcn_ $OUT/dcn_int8_time.jsonl; cat $OUT/ops_timing.jsonl
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "dcn" in r["Name"]: print(r["Name"][:110], r["Calls"], r["AverageNs"])
PY
