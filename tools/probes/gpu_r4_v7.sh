#!/bin/bash
# r04 visit 7: graph-replay timing of the dispatch candidates, few-row GEMM per layer, fused refinement, new dispatch table
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v7; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_geometry_gpu.py tests/test_small_gemm_gpu.py tests/test_tile_gemm_gpu.py tests/test_model_gpu.py tests/test_bevdet_gpu.py -q 2>&1 | tail -40 ) > $OUT/pytest_subset.log
( timeout 200 python tools/small_gemm_time.py 2>&1 | grep "^{" ) > $OUT/small_gemm_time.jsonl
( timeout 600 python tools/dump_dispatch.py $OUT/dispatch_gfx950.json 2>&1 | tail -3 ) > $OUT/dump_dispatch.log
cp $OUT/dispatch_gfx950.json bevformer_tensorrt_amd/dispatch_gfx950.json 2>/dev/null
( timeout 200 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{"
  timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{"
  timeout 200 python tools/model_bench.py small tiny --graph --frames 14 2>&1 | grep "^{" ) > $OUT/model_bench.jsonl
bash tools/model_profile.sh r4v7/model base > $OUT/model_frame_kernel_trace.txt 2>&1; rm -rf $OUT/model/prof
tail -25 $OUT/pytest_subset.log; cat $OUT/small_gemm_time.jsonl; cat $OUT/dump_dispatch.log; cat $OUT/model_bench.jsonl; head -24 $OUT/model_frame_kernel_trace.txt | cut -c1-130
