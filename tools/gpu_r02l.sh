#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02l; mkdir -p $OUT; export TMPDIR=/tmp
( MASKS=512,256,384,320,32,128,512,256,384 timeout 600 python tools/hm4_probe.py ablate 2>&1 | grep "{" ) > $OUT/hm4_variants.jsonl
( timeout 600 python -m pytest tests/test_mdconv_gpu.py -q -k "int8" 2>&1 | tail -5 ) > $OUT/pytest.log
cat $OUT/hm4_variants.jsonl; tail -3 $OUT/pytest.log
