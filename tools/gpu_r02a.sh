#!/bin/bash
# r02 visit A: new parity tests + experimental groups + baseline bench on this box
OUT=gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_geometry_gpu.py tests/test_wrappers_gpu.py tests/test_model_gpu.py -q -s 2>&1 | tail -60 ) > $OUT/pytest_new.log
( BEVOPS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_msda_int8_gpu.py tests/test_mdconv_gpu.py -q 2>&1 | tail -30 ) > $OUT/pytest_experimental.log
( timeout 600 python bench.py --steps 10 --warmup 2 --no-end-to-end 2>&1 | tail -1 ) > $OUT/bench.json
tail -25 $OUT/pytest_new.log; tail -8 $OUT/pytest_experimental.log; cat $OUT/bench.json
