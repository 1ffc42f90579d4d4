#!/bin/bash
# Targeted GPU check: selected test files + op timings.  usage: tools/gpu_quick.sh <tag> "<pytest args>"
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest $2 -q 2>&1 | tail -30 ) > $OUT/pytest.log
( timeout 300 python tools/ops_timing.py 2>&1 | grep "{" ) > $OUT/ops_timing.jsonl
tail -8 $OUT/pytest.log; cat $OUT/ops_timing.jsonl
