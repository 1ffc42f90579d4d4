#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest_all.log
( timeout 600 python tools/int8_model_delta.py tiny small --dense 2>&1 | grep "{" ) > $OUT/int8_model_delta.jsonl
( timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 ) > $OUT/bench.json
tail -12 $OUT/pytest_all.log; cat $OUT/int8_model_delta.jsonl; cat $OUT/bench.json
