OUT=gpurun_out/r01f
export TMPDIR=/tmp
rm -rf $OUT/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | tail -2 ) > $OUT/rocprof.log
tail -1 $OUT/rocprof.log | cut -c1-400
