#!/bin/bash
# r02 visit B: hm4 parity (fp16 + int8) and A/B timings
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_msda_hm4_gpu.py -q -x 2>&1 | tail -40 ) > $OUT/pytest_hm4.log
( timeout 900 python -m pytest tests/test_msda_hm4_gpu.py -q 2>&1 | tail -60 ) > $OUT/pytest_hm4_all.log
( timeout 600 python tools/hm4_probe.py 2>&1 | grep "{" ) > $OUT/hm4_probe.jsonl
( timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_msda_int8_gpu.py tests/test_msda_hm_gpu.py tests/test_full_size_gpu.py tests/test_sca_fused_gpu.py -q 2>&1 | tail -15 ) > $OUT/pytest_msda.log
tail -30 $OUT/pytest_hm4_all.log; cat $OUT/hm4_probe.jsonl; tail -5 $OUT/pytest_msda.log
