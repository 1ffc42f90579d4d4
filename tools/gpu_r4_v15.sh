#!/bin/bash
# r04 visit 15 (final code): full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the bench command
# (PMC passes and the INT8 frame trace: visit 10, tools/gpu_r4_v8.sh -- the default code paths are unchanged since)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v15; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu_tail.log
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $OUT/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 ) > $OUT/bench_n1.json
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; cat $OUT/pytest_gpu_tail.log | tail -12; tail -2 $OUT/smoke.log; cat $OUT/bench_n1.json; tail -3 $OUT/bench.err
