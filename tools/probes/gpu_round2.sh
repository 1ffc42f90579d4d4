#!/bin/bash
# One r02 evidence visit: tests, smoke, bench (fp16 + INT8 sub-record), rocprof kernel stats + PMC passes of the
# same command, SQ counters of the two SCA kernels, op timings, end-to-end model.  usage: tools/gpu_round2.sh <tag>
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1
nproc >> $OUT/rocminfo.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $OUT/pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -6 ) > $OUT/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > $OUT/bench.json
( timeout 300 python tools/ops_timing.py 2>&1 | grep "{" ) > $OUT/ops_timing.jsonl
( timeout 300 python tools/bevdet_slice.py 2>&1 | grep "{" ) > $OUT/bevdet_slice.jsonl
( timeout 600 python tools/hm4_probe.py 2>&1 | grep "{" ) > $OUT/hm4_probe.jsonl
( timeout 600 python tools/model_bench.py --graph 2>&1 | grep "{" ) > $OUT/model_bench.jsonl
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; tail -4 $OUT/pytest.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench.json; cat $OUT/model_bench.jsonl
