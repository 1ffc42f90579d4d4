#!/bin/bash
# One r03 evidence visit: tests, smoke, bench (end-to-end fp16 + INT8, hot path, rooflines), rocprof kernel stats +
# PMC passes of the hot-path command, hm5 probe, dense-layer / int8-layer / convolution timings, projected-SCA
# timing, model frames with the round's work off and on, fp16 and INT8 frame kernel traces.
# usage: tools/gpu_round3.sh <tag> [notests]
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1
nproc >> $OUT/rocminfo.txt
if [ "$2" != "notests" ]; then
  ( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log
  ( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -6 ) > $OUT/smoke.log
fi
( timeout 1200 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > $OUT/bench.json
( timeout 300 python tools/hm5_probe.py 16,1000,1001,1512 2>&1 | grep "{" ) > $OUT/hm5_probe.jsonl
( timeout 300 python tools/sca_projected_time.py 2>&1 | grep "{" ) > $OUT/sca_projected_time.jsonl
( timeout 300 python tools/dense_time.py 2>&1 | grep "{" ) > $OUT/dense_time.jsonl
( timeout 300 python tools/linear_q_time.py 2>&1 | grep "{" ) > $OUT/linear_q_time.jsonl
( timeout 300 python tools/conv_time.py 2>&1 | grep "{" ) > $OUT/conv_time.jsonl
( timeout 600 python tools/model_bench.py --graph 2>&1 | grep "{" ) > $OUT/model_bench.jsonl
# the frame with the round-3 work switched off / its measured dispatch switched off / on, same box back to back
( BEVOPS_R3_FUSIONS=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": false, /'
  BEVOPS_DENSE_TUNE=0 timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, "measured_dispatch": false, /'
  timeout 300 python tools/model_bench.py base --graph --frames 14 2>&1 | grep "{" | sed 's/^{/{"r3_fusions": true, "measured_dispatch": true, /'
  BEVOPS_FUSED_QUANT=0 timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": false, /'
  timeout 600 python tools/model_bench.py base --graph --int8 --frames 14 2>&1 | grep "{" | sed 's/^{/{"fused_quant": true, /' ) > $OUT/model_bench_r3_ab.jsonl
bash tools/model_profile.sh $TAG/model base > $OUT/model_frame_kernel_trace.txt 2>&1
bash tools/model_profile.sh $TAG/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; tail -4 $OUT/pytest.log 2>/dev/null; tail -2 $OUT/smoke.log 2>/dev/null; cat $OUT/bench.json; cat $OUT/model_bench.jsonl; cat $OUT/model_bench_r3_ab.jsonl
