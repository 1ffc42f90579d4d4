#!/bin/bash
# r05 visit 22: diagnostic of the model-level plan / no-plan difference, second step (+ the fixed int8 stem test)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v22; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/probes/plan_model_diff2.py > $OUT/diff2.log 2>&1; grep -v amdgpu.ids $OUT/diff2.log | tail -12
( timeout 200 python -m pytest -q -p no:cacheprovider -x -m gpu tests/test_stem_gpu.py -k "int8 or backbone or validates" 2>&1 | tail -5 ) > $OUT/tests_stem.log; tail -3 $OUT/tests_stem.log
