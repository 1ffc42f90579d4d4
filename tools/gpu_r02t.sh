#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02t; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_camera_shard_gpu.py -q 2>&1 | tail -30 ) > $OUT/pytest.log
( timeout 300 python tools/rccl_two_ranks_one_gpu.py 2>&1 | tail -12 ) > $OUT/two_ranks.log
cat $OUT/pytest.log; cat $OUT/two_ranks.log
