#!/bin/bash
# Same-box interleaved A/B of the base frame between two environments (HIP-graph replay, per-frame-synchronised
# protocol, three rounds):   gpurun -- 'bash tools/frame_env_ab.sh TAG "VAR=a" "VAR=b" [model_bench args]'
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT; A=$2; B=$3; shift 3
CMD="python tools/model_bench.py ${@:-base} --graph --no-clone --static-image --frames 40"
for i in 1 2 3; do
  echo "{\"env\": \"$A\"}"; env $A $CMD 2>/dev/null | tail -1
  echo "{\"env\": \"$B\"}"; env $B $CMD 2>/dev/null | tail -1
done > $OUT/frame_env_ab.jsonl
cut -c1-200 $OUT/frame_env_ab.jsonl
