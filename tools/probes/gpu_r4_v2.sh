#!/bin/bash
# r04 visit 2: the int8 activation chain -- operator tests, regression of the touched kernels, engine A/B, attribution
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v2; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_int8_chain_gpu.py -q -x 2>&1 | tail -25 ) > $OUT/pytest_chain.log
( timeout 900 python -m pytest tests/test_tile_gemm_gpu.py tests/test_linear_q_gpu.py tests/test_mdconv_gpu.py tests/test_epilogue_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -8 ) > $OUT/pytest_regress.log
( timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{" 
  timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{" | sed 's/^{/{"chain": true, /'
  BEVOPS_INT8_CHAIN=0 timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{" | sed 's/^{/{"chain": false, /' ) > $OUT/model_bench.jsonl 2> $OUT/model_bench.err
( timeout 600 python tools/int8_attribution.py base --calib 16 --frames 3 --chain --no-fp32 2>&1 | grep "^{" ) > $OUT/int8_attribution_chain.jsonl
bash tools/model_profile.sh r4v2/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1
rm -rf $OUT/model_int8/prof
cat $OUT/pytest_chain.log; cat $OUT/pytest_regress.log; cat $OUT/model_bench.jsonl; tail -3 $OUT/model_bench.err; cut -c1-330 $OUT/int8_attribution_chain.jsonl; head -40 $OUT/model_frame_int8_kernel_trace.txt | cut -c1-150
