#!/bin/bash
# r05 visit 8: offset convolution with the K split across three waves per tile
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v8; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_mdconv_gpu.py -q -p no:cacheprovider -x -k "conv_offset or nhwc or packed" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 300 python tools/conv_offset_time.py > $OUT/conv_offset_time.jsonl 2> $OUT/err.log; cat $OUT/conv_offset_time.jsonl; tail -3 $OUT/err.log
timeout 600 python tools/model_bench.py base --graph --frames 30 > $OUT/model_bench.jsonl 2>> $OUT/err.log; cat $OUT/model_bench.jsonl
