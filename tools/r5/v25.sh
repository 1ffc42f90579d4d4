#!/bin/bash
# r05 visit 25: library dense layers at shifted operand addresses; frames with the framework addmm excluded
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v25; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python tools/probes/dense_alignment.py > $OUT/dense_alignment.log 2>&1; grep -v amdgpu.ids $OUT/dense_alignment.log | tail -8 | cut -c1-500
