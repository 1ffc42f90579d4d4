#!/bin/bash
# Mid-round visit: new model tests, fp16 frame A/B of the launch-count work, kernel traces of the fp16 and INT8 frames.
TAG=${1:-r3p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -8 ) > $OUT/pytest.log
( for i in 1 2; do BEVOPS_R3_FUSIONS=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": false, /'; timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, /'; done ) > $OUT/model_bench_r3_ab.jsonl
( timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" ) > $OUT/model_bench_int8.jsonl
bash tools/model_profile.sh $TAG/model base > $OUT/model_frame_kernel_trace.txt 2>&1
bash tools/model_profile.sh $TAG/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1
( timeout 300 python tools/model_ops_profile.py base 60 2>&1 | tail -70 ) > $OUT/model_ops_profile.txt
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
tail -5 $OUT/pytest.log; cat $OUT/model_bench_r3_ab.jsonl $OUT/model_bench_int8.jsonl; head -30 $OUT/model_frame_int8_kernel_trace.txt
