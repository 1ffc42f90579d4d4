#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mdconv_gpu.py tests/test_linear_q_gpu.py tests/test_image_gpu.py -q 2>&1 | tail -40 ) > $OUT/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python $GRAFT_REPO_ROOT/tools/hm4_probe.py kernels > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -30 $OUT/pytest.log; grep dcn_int8 $OUT/prof.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "dcn" in r["Name"] or "im2col" in r["Name"] or "gemm_tn_s8" in r["Name"]: print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
