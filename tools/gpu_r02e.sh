#!/bin/bash
OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_msda_hm4_gpu.py tests/test_msda_int8_gpu.py -q 2>&1 | tail -30 ) > $OUT/pytest_hm4.log
( timeout 600 python tools/hm4_probe.py 2>&1 | grep "{" ) > $OUT/hm4_probe.jsonl
( timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 ) > $OUT/bench.json
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $OUT/pytest_all.log
tail -8 $OUT/pytest_hm4.log; cat $OUT/hm4_probe.jsonl; cat $OUT/bench.json; tail -8 $OUT/pytest_all.log
