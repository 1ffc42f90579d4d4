#!/bin/bash
# r05 visit 20: the one-kernel stem -- its tests, the GPU test files visit 19 did not reach, its timing against the
# two-pass form, and the base frame with / without it (interleaved)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v20; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest -q -p no:cacheprovider -x -m gpu tests/test_stem_gpu.py 2>&1 | tail -15 ) > $OUT/tests_stem.log
tail -4 $OUT/tests_stem.log
timeout 200 python tools/stem_time.py 2>> $OUT/err.log > $OUT/stem_time.jsonl; cat $OUT/stem_time.jsonl
( timeout 900 python -m pytest -q -p no:cacheprovider -m gpu tests/test_tile_gemm_gpu.py tests/test_int8_chain_gpu.py \
    tests/test_camera_shard_gpu.py tests/test_model_gpu.py tests/test_bevdet_gpu.py 2>&1 | tail -15 ) > $OUT/tests_rest.log
tail -4 $OUT/tests_rest.log
for r in 1 2; do
  BEVOPS_STEM_FUSED=0 timeout 300 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"stem\": \"library convolution + pooling pass\", /" >> $OUT/model_bench_stem.jsonl
  timeout 300 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"stem\": \"one kernel\", /" >> $OUT/model_bench_stem.jsonl
done
cut -c1-260 $OUT/model_bench_stem.jsonl; tail -5 $OUT/err.log
