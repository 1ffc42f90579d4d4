#!/bin/bash
# r05 visit 19: GPU tests of the files the dead-branch clean-up touched + the planned SCA's direct stores / unrolled
# reduce, then their interleaved A/B on the rig geometry
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v19; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest -q -p no:cacheprovider -x -m gpu tests/test_sca_fused_gpu.py tests/test_msda_hm5_gpu.py \
    tests/test_msda_hm4_gpu.py tests/test_msda_hm_gpu.py tests/test_msda_int8_gpu.py tests/test_tile_gemm_gpu.py \
    tests/test_int8_chain_gpu.py tests/test_camera_shard_gpu.py tests/test_model_gpu.py tests/test_msda_gpu.py 2>&1 | tail -15 ) > $OUT/tests.log
for r in 1 2; do timeout 300 python tools/sca_frame_time.py --ks 2 2>> $OUT/err.log >> $OUT/sca_direct_ab.jsonl; done
tail -5 $OUT/tests.log; cat $OUT/sca_direct_ab.jsonl; tail -5 $OUT/err.log
