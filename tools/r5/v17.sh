#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v17; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2; do for o in 0 1; do
  BEVOPS_IMAGE_NHWC=$o timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"image_nhwc\": $o, /" >> $OUT/model_bench_image_nhwc.jsonl
done; done
cat $OUT/model_bench_image_nhwc.jsonl
