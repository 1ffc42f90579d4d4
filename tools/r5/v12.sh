#!/bin/bash
# r05 visit 12: TSA on the layout-preserving kernel (BEVOPS_TSA_LOCAL) inside the graph-replayed frame, interleaved
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v12; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2 3; do for o in 0 1; do
  BEVOPS_TSA_LOCAL=$o timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image 2>> $OUT/err.log | sed "s/^{/{\"tsa_local\": $o, /" >> $OUT/model_bench_tsa_local.jsonl
done; done
cat $OUT/model_bench_tsa_local.jsonl
