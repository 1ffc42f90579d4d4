#!/bin/bash
# r04 visit 8 (final code): full GPU suite, smoke, the default bench line, rocprofv3 kernel stats + PMC passes, INT8 frame trace
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v8; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $OUT/pytest_gpu_tail.log
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $OUT/smoke.log
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 ) > $OUT/bench_n1.json
bash tools/model_profile.sh r4v8/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1; rm -rf $OUT/model_int8/prof
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
( timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; cat $OUT/pytest_gpu_tail.log | tail -12; tail -2 $OUT/smoke.log; cat $OUT/bench_n1.json; tail -3 $OUT/bench.err; head -16 $OUT/model_frame_int8_kernel_trace.txt | cut -c1-130
