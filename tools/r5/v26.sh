#!/bin/bash
# r05 visit 26: which operator call of the eager base frame is not run-to-run deterministic
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v26; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/probes/op_determinism.py > $OUT/op_determinism.log 2>&1; grep -v amdgpu.ids $OUT/op_determinism.log | tail -10 | cut -c1-400
