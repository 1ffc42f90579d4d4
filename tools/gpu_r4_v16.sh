#!/bin/bash
# r04 visit 16: the full GPU suite again, verbose log kept (visit 15's run died with a segmentation fault early on)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v16; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -X faulthandler -m pytest tests -m gpu -v -p no:cacheprovider > $OUT/pytest_gpu_full.log 2>&1
echo "rc=$?" >> $OUT/pytest_gpu_full.log
grep -n "Fatal\|Segmentation\|INTERNALERROR\|passed\|failed\|rc=" $OUT/pytest_gpu_full.log | head -20
grep -n "PASSED\|FAILED\|ERROR" $OUT/pytest_gpu_full.log | tail -3
