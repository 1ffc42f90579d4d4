#!/bin/bash
# r05 visit 16: channels-last static image buffer: image / model / shard / bench tests, frame time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v16; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_image_gpu.py tests/test_model_gpu.py tests/test_geometry_gpu.py tests/test_camera_shard_gpu.py tests/test_bevdet_gpu.py -q -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for r in 1 2; do timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image >> $OUT/model_bench.jsonl 2>> $OUT/err.log; done
timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image --int8 >> $OUT/model_bench.jsonl 2>> $OUT/err.log; cat $OUT/model_bench.jsonl; tail -3 $OUT/err.log
