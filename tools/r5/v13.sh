#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v13; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/glue_ops.py base > $OUT/glue_ops.txt 2> $OUT/err.log; cat $OUT/glue_ops.txt | cut -c1-160; tail -5 $OUT/err.log
