#!/bin/bash
# r04 visit 1 (diagnostics on the round-3 engine): per-site INT8 error attribution, operator-level profile of the INT8 frame
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v1; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python tools/int8_attribution.py base --calib 16 --frames 3 2>&1 | grep "^{" ) > $OUT/int8_attribution.jsonl
( timeout 300 python tools/model_ops_profile.py base 90 --int8 2>&1 | tail -95 ) > $OUT/ops_profile_int8.txt
( timeout 300 python tools/model_ops_profile.py base 60 2>&1 | tail -65 ) > $OUT/ops_profile_fp16.txt
cat $OUT/int8_attribution.jsonl | cut -c1-400; head -50 $OUT/ops_profile_int8.txt | cut -c1-180
