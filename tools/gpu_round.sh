#!/bin/bash
# One GPU visit: tests, smoke, bench, rocprof (kernel stats + PMC), end-to-end model.  Usage: tools/gpu_round.sh <tag> [sweep]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1
nproc >> $OUT/rocminfo.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 ) > $OUT/pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -10 ) > $OUT/smoke.log
if [ -n "$2" ]; then ( VARIANTS=0 timeout 900 python tools/msda_sweep.py 2>&1 ) > $OUT/sweep.jsonl; fi
( timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 ) > $OUT/bench.json
( timeout 300 python tools/dcn_time.py 0 2 1 2>&1 | grep shape ) > $OUT/dcn_time.jsonl
( timeout 600 python tools/model_bench.py --graph 2>&1 | grep "{" ) > $OUT/model_bench.jsonl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | tail -5 ) > $OUT/rocprof.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | tail -3 ) > $OUT/rocprof_pmc_fetch.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | tail -3 ) > $OUT/rocprof_pmc_write.log
( timeout 300 bash tools/model_profile.sh $TAG/trace base 2>&1 | tail -45 ) > $OUT/model_trace.txt
rm -rf $OUT/trace/prof
find $OUT -name "*.csv" | head -20; du -sh $OUT
tail -5 $OUT/pytest.log; cat $OUT/smoke.log; cat $OUT/bench.json; cat $OUT/model_bench.jsonl
