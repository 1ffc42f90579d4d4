#!/bin/bash
# Re-verification of the final state after tools/gpu_round3.sh r03b (whose probes / rocprof passes cover kernels that
# did not change afterwards): the GPU suite, smoke, the bench line, the frame A/B runs and the INT8 error budget of
# the bench build at base.  usage: tools/gpu_round3_final.sh <tag>
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -6 ) > $OUT/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > $OUT/bench.json
( BEVOPS_R3_FUSIONS=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": false, /'
  BEVOPS_DENSE_TUNE=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, "measured_dispatch": false, /'
  timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, "measured_dispatch": true, /'
  BEVOPS_FUSED_QUANT=0 timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": false, /'
  timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": true, /' ) > $OUT/model_bench_r3_ab.jsonl
( timeout 600 python tools/int8_model_delta.py base --engine --calib 16 --frames 3 2>&1 | grep "{" ) > $OUT/int8_model_delta.jsonl
tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log; cat $OUT/bench.json $OUT/model_bench_r3_ab.jsonl $OUT/int8_model_delta.jsonl
