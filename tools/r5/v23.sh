#!/bin/bash
# r05 visit 23: which module of the eager base frame is not run-to-run deterministic
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v23; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/probes/frame_determinism.py base > $OUT/determinism.log 2>&1; grep -v amdgpu.ids $OUT/determinism.log | tail -20
