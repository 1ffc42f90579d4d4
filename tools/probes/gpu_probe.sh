#!/bin/bash
# usage: TAG=x VARS="16,1000,.." DISTS=uniform,rig [TESTS="pytest args"] bash tools/gpu_probe.sh
OUT=gpurun_out/${TAG:-probe}; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then ( timeout 900 python -m pytest $TESTS -x -q 2>&1 | tail -15 ) > $OUT/pytest.log; cat $OUT/pytest.log; fi
( timeout 600 python tools/hm5_probe.py "$VARS" "${DISTS:-uniform,rig}" ${ROUNDS:-3} 2>&1 | grep "{" ) > $OUT/hm5_probe.jsonl
cat $OUT/hm5_probe.jsonl
