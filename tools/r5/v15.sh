#!/bin/bash
# r05 visit 15: TSA glue kernels (bit identity), model tests, frame time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v15; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_epilogue_gpu.py tests/test_model_gpu.py tests/test_wrappers_gpu.py tests/test_int8_chain_gpu.py -q -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for r in 1 2; do timeout 400 python tools/model_bench.py base --graph --frames 40 --no-clone --static-image >> $OUT/model_bench.jsonl 2>> $OUT/err.log; done; cat $OUT/model_bench.jsonl
