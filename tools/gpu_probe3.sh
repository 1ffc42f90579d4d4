#!/bin/bash
# INT8 dense-layer visit: tests of the fused-quantise GEMM, per-layer timings, INT8 frame A/B and kernel trace.
TAG=${1:-r3q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_linear_q_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -8 ) > $OUT/pytest.log
( timeout 300 python tools/linear_q_time.py 2>&1 | grep "{" ) > $OUT/linear_q_time.jsonl
( BEVOPS_FUSED_QUANT=0 timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": false, /' ) > $OUT/model_bench_int8.jsonl
( timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": true, /' ) >> $OUT/model_bench_int8.jsonl
( timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" ) >> $OUT/model_bench_int8.jsonl
bash tools/model_profile.sh $TAG/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
tail -5 $OUT/pytest.log; cat $OUT/linear_q_time.jsonl $OUT/model_bench_int8.jsonl; head -24 $OUT/model_frame_int8_kernel_trace.txt
