#!/bin/bash
# r04 visit 9: INT8 FPN convolutions on a pre-quantised input: test, frame A/B, attribution of the engine with them
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v9; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_int8_chain_gpu.py -q 2>&1 | tail -15 ) > $OUT/pytest_subset.log
( BEVOPS_INT8_FPN=0 timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{" | sed 's/^{/{"int8_fpn": false, /'
  timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{" | sed 's/^{/{"int8_fpn": true, /'
  timeout 200 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{" ) > $OUT/model_bench.jsonl
( timeout 400 python tools/int8_attribution.py base --calib 16 --frames 3 --chain --no-fp32 2>&1 | grep "^{" ) > $OUT/int8_attribution.jsonl
bash tools/model_profile.sh r4v9/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1; rm -rf $OUT/model_int8/prof
tail -8 $OUT/pytest_subset.log; cat $OUT/model_bench.jsonl; head -6 $OUT/int8_attribution.jsonl | cut -c1-330; grep -E "conv.neck|all but conv" $OUT/int8_attribution.jsonl | cut -c1-330; head -14 $OUT/model_frame_int8_kernel_trace.txt | cut -c1-130
