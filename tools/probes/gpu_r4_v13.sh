#!/bin/bash
# r04 visit 13: wide k-steps of the int8 chain GEMMs: bits, time
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v13; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_int8_chain_gpu.py -q -k "linear_int8_chain" 2>&1 | tail -8 ) > $OUT/pytest_chain.log
( timeout 300 python tools/tile_wide_ab.py 2>&1 | grep "^{" ) > $OUT/tile_wide_ab.jsonl
cat $OUT/pytest_chain.log; cat $OUT/tile_wide_ab.jsonl
