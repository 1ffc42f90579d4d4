#!/bin/bash
# r04 visit 5: sharded graph capture, fast INT8 DCNv2 flavour (tests, accuracy, speed), dispatch table, per-frame host cost
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v5; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_camera_shard_gpu.py tests/test_int8_chain_gpu.py tests/test_bevdet_gpu.py tests/test_model_gpu.py tests/test_tile_gemm_gpu.py -q 2>&1 | tail -60 ) > $OUT/pytest_subset.log
( timeout 400 python tools/dump_dispatch.py $OUT/dispatch_gfx950.json 2>&1 | tail -3 ) > $OUT/dump_dispatch.log
( timeout 150 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "^{"
  timeout 150 python tools/model_bench.py base --graph --frames 14 --no-clone 2>&1 | grep "^{"
  timeout 150 python tools/model_bench.py base --graph --frames 14 --static-image 2>&1 | grep "^{"
  timeout 150 python tools/model_bench.py base --graph --frames 14 --no-clone --static-image 2>&1 | grep "^{"
  timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "^{" ) > $OUT/model_bench.jsonl
( timeout 400 python tools/int8_attribution.py base --calib 16 --frames 3 --chain --no-fp32 2>&1 | grep "^{" | head -4 ) > $OUT/int8_attribution_fastdcn.jsonl
bash tools/model_profile.sh r4v5/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1; rm -rf $OUT/model_int8/prof
tail -40 $OUT/pytest_subset.log; cat $OUT/dump_dispatch.log; cat $OUT/model_bench.jsonl; cut -c1-330 $OUT/int8_attribution_fastdcn.jsonl; head -14 $OUT/model_frame_int8_kernel_trace.txt | cut -c1-140
