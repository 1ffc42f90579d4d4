#!/bin/bash
# r04 visit 6: few-row GEMM (tests, per-layer timing, frame), rotate wide stores, sharded graph, regenerated dispatch table
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v6; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 700 python -m pytest tests/test_small_gemm_gpu.py tests/test_sampler_gpu.py tests/test_camera_shard_gpu.py tests/test_tile_gemm_gpu.py tests/test_model_gpu.py -q 2>&1 | tail -40 ) > $OUT/pytest_subset.log
( timeout 300 python tools/dense_time.py 2>&1 | grep "^{" ) > $OUT/dense_time.jsonl
( timeout 200 python tools/ops_timing.py 2>&1 | grep "^{" ) > $OUT/ops_timing.jsonl
( timeout 500 python tools/dump_dispatch.py $OUT/dispatch_gfx950.json 2>&1 | tail -3 ) > $OUT/dump_dispatch.log
( timeout 200 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{"
  timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{"
  BEVOPS_DENSE_TUNE=0 timeout 200 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{" | sed 's/^{/{"dense_tune": 0, /' ) > $OUT/model_bench.jsonl
tail -25 $OUT/pytest_subset.log; tail -8 $OUT/dense_time.jsonl; grep rotate $OUT/ops_timing.jsonl; cat $OUT/dump_dispatch.log; cat $OUT/model_bench.jsonl
