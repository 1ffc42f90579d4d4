#!/bin/bash
# r05 visit 6: the whole GPU suite on the pruned library
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v6; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
