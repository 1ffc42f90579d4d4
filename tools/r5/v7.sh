#!/bin/bash
# r05 visit 7: bench.py's sharded path on one rank (GPU test); fp16 frame kernel trace with the planned SCA sampling
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -X faulthandler -m pytest tests/test_bench_gpu.py -v -p no:cacheprovider > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
grep -n "passed\|failed\|FAILED\|Error\|rc=" $OUT/tests.log | head; tail -30 $OUT/tests.log | grep -v "^$" | tail -15
bash tools/model_profile.sh r5v7/model base > $OUT/model_frame_kernel_trace.txt 2>&1; rm -rf $OUT/model/prof
head -40 $OUT/model_frame_kernel_trace.txt | cut -c1-140
