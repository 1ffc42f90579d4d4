#!/bin/bash
# r05 visit 28: what the reproducible dispatch (BEVOPS_DENSE_TUNE=0: dense layers and convolutions on the hand-written
# kernels, bit-reproducible frames) costs against the measured default, base frame, interleaved
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v28; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"dispatch\": \"measured default\", /" >> $OUT/model_bench_reproducible.jsonl
BEVOPS_DENSE_TUNE=0 timeout 200 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"dispatch\": \"reproducible (BEVOPS_DENSE_TUNE=0)\", /" >> $OUT/model_bench_reproducible.jsonl
cut -c1-240 $OUT/model_bench_reproducible.jsonl; tail -2 $OUT/err.log
