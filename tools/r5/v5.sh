#!/bin/bash
# r05 visit 5: the "scatter" exchange on RCCL (one rank) incl. graph capture; Amdahl table with the measured scatter frame
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v5; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_camera_shard_gpu.py tests/test_sca_fused_gpu.py tests/test_model_gpu.py -v -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
grep -n "passed\|failed\|FAILED\|Error\|rc=" $OUT/tests.log | head -20
timeout 900 python tools/shard_amdahl.py base > $OUT/shard_amdahl.jsonl 2> $OUT/amdahl.err; echo "amdahl rc=$?"; cat $OUT/shard_amdahl.jsonl; tail -5 $OUT/amdahl.err
