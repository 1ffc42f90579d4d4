#!/bin/bash
# round 3, first GPU visit: hm5 parity + probe + per-kernel trace
OUT=gpurun_out/${TAG:-r3b}; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_msda_hm5_gpu.py -x -q 2>&1 | tail -25 ) > $OUT/pytest_hm5.log
( timeout 600 python -m pytest tests/test_msda_int8_gpu.py tests/test_ref_kernels_gpu.py tests/test_ref_kernels_live_gpu.py tests/test_msda_hm_gpu.py -q -k "int8 or hm3 or hm_staged" 2>&1 | tail -12 ) > $OUT/pytest_int8.log
( timeout 600 python tools/hm5_probe.py 2>&1 | grep "{" ) > $OUT/hm5_probe.jsonl
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/tools/hm5_probe.py 16,1000,1064 uniform,rig 1 > /dev/null 2>&1 )
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' > $OUT/kernel_stats_head.txt
tail -25 $OUT/pytest_hm5.log; cat $OUT/pytest_int8.log; cat $OUT/hm5_probe.jsonl; cat $OUT/kernel_stats_head.txt
