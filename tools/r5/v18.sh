#!/bin/bash
# r05 visit 18: the round-4 behaviour (switches) against the round-5 defaults, interleaved on one box, fp16 base frame
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v18; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2 3; do
  BEVOPS_SCA_PLAN=0 BEVOPS_TSA_LOCAL=0 BEVOPS_TSA_GLUE=0 timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image --conv-variant 2 2>> $OUT/err.log | sed "s/^{/{\"build\": \"round-4 behaviour\", /" >> $OUT/model_bench_r4_vs_r5.jsonl
  timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"build\": \"round-5 defaults\", /" >> $OUT/model_bench_r4_vs_r5.jsonl
done
cat $OUT/model_bench_r4_vs_r5.jsonl
