#!/bin/bash
# r05 visit 21: diagnostic of the model-level plan / no-plan difference
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v21; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/probes/plan_model_diff.py > $OUT/diff.log 2>&1; tail -30 $OUT/diff.log
