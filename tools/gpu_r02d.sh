#!/bin/bash
OUT=gpurun_out/r02d; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python tools/int8_diag.py 2>&1 | tail -80 ) > $OUT/int8_diag.txt
cat $OUT/int8_diag.txt
