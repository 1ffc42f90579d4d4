#!/bin/bash
# r04 visit 11: TSA's MSDA on the layout-preserving kernel (the BEV grid has locality): probe, test, frame A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v11; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 200 python tools/tsa_geometry_probe.py 2>&1 | grep "^{" ) > $OUT/tsa_geometry_probe.jsonl
( timeout 300 python -m pytest tests/test_msda_gpu.py -q -k "local_entry or golden" 2>&1 | tail -6 ) > $OUT/pytest_subset.log
( BEVOPS_TSA_LOCAL=0 timeout 200 python tools/model_bench.py base small --graph --frames 14 2>&1 | grep "^{" | sed 's/^{/{"tsa_local": false, /'
  BEVOPS_TSA_LOCAL=1 timeout 200 python tools/model_bench.py base small --graph --frames 14 2>&1 | grep "^{" | sed 's/^{/{"tsa_local": true, /' ) > $OUT/model_bench.jsonl
cat $OUT/tsa_geometry_probe.jsonl; tail -3 $OUT/pytest_subset.log; cat $OUT/model_bench.jsonl
