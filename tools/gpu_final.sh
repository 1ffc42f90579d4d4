#!/bin/bash
# Last visit of a round: the suite exactly as the driver runs it, smoke, and the INT8 hot-path bench line.
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $OUT/pytest.log
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $OUT/smoke.log
( timeout 300 python bench.py --dtype int8 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 ) > $OUT/bench_int8.json
cat $OUT/pytest.log $OUT/smoke.log $OUT/bench_int8.json
