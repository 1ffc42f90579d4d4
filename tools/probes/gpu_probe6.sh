#!/bin/bash
# INT8 convolution visit: tests, INT8 frame, INT8 error budget of the bench build at base, small / tiny frame A/B.
TAG=${1:-r3i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_linear_q_gpu.py tests/test_tile_gemm_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -8 ) > $OUT/pytest.log
( timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" ) > $OUT/model_bench_int8.jsonl
( timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" ) >> $OUT/model_bench_int8.jsonl
( timeout 900 python tools/int8_model_delta.py base --engine --calib 16 --frames 3 2>&1 | grep "{" ) > $OUT/int8_model_delta.jsonl
( for m in tiny small; do BEVOPS_R3_FUSIONS=0 timeout 300 python tools/model_bench.py $m --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": false, /'; timeout 300 python tools/model_bench.py $m --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, /'; done ) > $OUT/model_bench_small_ab.jsonl
tail -5 $OUT/pytest.log; cat $OUT/model_bench_int8.jsonl $OUT/int8_model_delta.jsonl $OUT/model_bench_small_ab.jsonl
