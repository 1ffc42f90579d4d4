#!/bin/bash
# Same-box interleaved A/B of the base frame (HIP-graph replay, per-frame-synchronised protocol), three rounds:
#   fp16 (default): round-5 kernels (DCNv2 wave order 13, planned SCA sampler 3015) against the round-6 defaults;
#   int8          : the INT8 engine with the round 2-5 DCNv2 wave order (13) against the default.
#   gpurun -- 'bash tools/frame_ab.sh TAG [int8]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT
B="python tools/model_bench.py base --graph --no-clone --static-image --frames 40"
if [ "$2" = int8 ]; then B="$B --int8"; OLD="--mdconv-variant 13"; F=frame_ab_int8.jsonl; else OLD="--mdconv-variant 13 --msda-variant 3015"; F=frame_ab.jsonl; fi
for i in 1 2 3; do
  $B $OLD 2>/dev/null | tail -1
  $B 2>/dev/null | tail -1
done > $OUT/$F
cat $OUT/$F | cut -c1-260
