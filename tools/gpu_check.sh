#!/bin/bash
# Short GPU visit: the whole GPU suite (no -x, every failure listed), the reference-kernel
# checker on this host, end-to-end frames/s (fused-linear A/B) and an operator-level profile.
# usage: tools/gpu_check.sh <tag> [skip-profile]
TAG=${1:-chk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $OUT/pytest.log
( timeout 300 python -m pytest tests/test_ref_kernels_cpu.py -q 2>&1 | tail -5 ) > $OUT/pytest_refk_cpu.log
( timeout 600 python tools/model_bench.py base --graph 2>&1 | grep "{" ) > $OUT/model_bench.jsonl
( BEVOPS_FUSED_LINEAR=0 timeout 600 python tools/model_bench.py base --graph 2>&1 | grep "{" | sed 's/^/two-launch: /' ) >> $OUT/model_bench.jsonl
if [ -z "$2" ]; then
  ( timeout 600 python tools/model_ops_profile.py base 80 2>&1 | tail -100 ) > $OUT/model_ops.txt
fi
tail -15 $OUT/pytest.log; cat $OUT/pytest_refk_cpu.log; cat $OUT/model_bench.jsonl; cat $OUT/model_ops.txt | cut -c1-250
