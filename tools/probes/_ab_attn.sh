OUT=$GRAFT_REPO_ROOT/gpurun_out/r6v27; mkdir -p $OUT
B="python tools/model_bench.py base --graph --no-clone --static-image --frames 40"
for i in 1 2 3; do
  BEVOPS_OWN_ATTN=0 $B 2>/dev/null | tail -1 | sed 's/}$/, "own_attn": 0}/'
  $B 2>/dev/null | tail -1 | sed 's/}$/, "own_attn": 1}/'
done > $OUT/frame_ab_attn.jsonl
for m in tiny small; do BEVOPS_OWN_ATTN=0 python tools/model_bench.py $m --graph --no-clone --static-image --frames 40 2>/dev/null | tail -1 | sed 's/}$/, "own_attn": 0}/'; python tools/model_bench.py $m --graph --no-clone --static-image --frames 40 2>/dev/null | tail -1 | sed 's/}$/, "own_attn": 1}/'; done >> $OUT/frame_ab_attn.jsonl
cat $OUT/frame_ab_attn.jsonl | cut -c1-300
