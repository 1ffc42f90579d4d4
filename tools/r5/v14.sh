#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v14; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/glue_copies.py > $OUT/glue_copies.txt 2> $OUT/err.log; cat $OUT/glue_copies.txt | cut -c1-160; tail -5 $OUT/err.log
