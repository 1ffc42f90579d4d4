#!/bin/bash
# Same-box interleaved A/B of the base frame (HIP-graph replay, per-frame-synchronised protocol): round-5 kernels
# (DCNv2 wave order 13, planned SCA sampler 3015) against the round-6 defaults, three rounds.
#   gpurun -- 'bash tools/frame_ab.sh TAG'
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT
B="python tools/model_bench.py base --graph --no-clone --static-image --frames 40"
for i in 1 2 3; do
  $B --mdconv-variant 13 --msda-variant 3015 2>/dev/null | tail -1
  $B 2>/dev/null | tail -1
done > $OUT/frame_ab.jsonl
cat $OUT/frame_ab.jsonl | cut -c1-260
