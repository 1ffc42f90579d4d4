#!/bin/bash
# conv3x3 visit: tests, conv timings, fp16 frame A/B, INT8 frame, kernel trace.
TAG=${1:-r3c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_tile_gemm_gpu.py tests/test_linear_q_gpu.py tests/test_epilogue_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -8 ) > $OUT/pytest.log
( timeout 300 python tools/conv_time.py 2>&1 | grep "{" ) > $OUT/conv_time.jsonl
( timeout 300 python tools/dense_time.py 2>&1 | grep "{" ) > $OUT/dense_time.jsonl
( for i in 1 2; do BEVOPS_DENSE_TUNE=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"measured_dispatch": false, /'; timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"measured_dispatch": true, /'; done ) > $OUT/model_bench_dispatch_ab.jsonl
( timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" ) > $OUT/model_bench_int8.jsonl
bash tools/model_profile.sh $TAG/model base > $OUT/model_frame_kernel_trace.txt 2>&1
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
tail -5 $OUT/pytest.log; cat $OUT/dense_time.jsonl $OUT/conv_time.jsonl $OUT/model_bench_dispatch_ab.jsonl $OUT/model_bench_int8.jsonl; head -30 $OUT/model_frame_kernel_trace.txt
