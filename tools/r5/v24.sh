#!/bin/bash
# r05 visit 24: which dense-layer candidate is not run-to-run deterministic; frames under the reproducible dispatch
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v24; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/probes/dense_determinism.py > $OUT/dense_determinism.log 2>&1; grep -v amdgpu.ids $OUT/dense_determinism.log | tail -12 | cut -c1-400
