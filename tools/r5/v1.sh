#!/bin/bash
# r05 visit 1: the two staged A-resident GEMMs run on a device for the first time (gated tests, then the A/B tools)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v1; mkdir -p $OUT; export TMPDIR=/tmp
BEVOPS_STAGED_TESTS=1 timeout 300 python -X faulthandler -m pytest tests -m gpu -k "a_resident" -v -p no:cacheprovider > $OUT/staged_tests.log 2>&1
echo "rc=$?" >> $OUT/staged_tests.log
grep -c PASSED $OUT/staged_tests.log; grep -n "FAILED\|rc=\|passed\|failed\|Fatal" $OUT/staged_tests.log | head -30
timeout 200 python tools/tsgemm_s8_ab.py --ares > $OUT/tsgemm_s8_ares_ab.jsonl 2> $OUT/tsgemm_s8_ares_ab.err; echo "s8 ab rc=$?"
cat $OUT/tsgemm_s8_ares_ab.jsonl; tail -3 $OUT/tsgemm_s8_ares_ab.err
timeout 200 python tools/tsgemm_f16_ares_ab.py > $OUT/tsgemm_f16_ares_ab.jsonl 2> $OUT/tsgemm_f16_ares_ab.err; echo "f16 ab rc=$?"
cat $OUT/tsgemm_f16_ares_ab.jsonl; tail -3 $OUT/tsgemm_f16_ares_ab.err
