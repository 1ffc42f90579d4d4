#!/bin/bash
# r05 visit 2: the planned fused SCA sampling (bit identity with the chunked kernel, timing, per-kernel times)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_sca_fused_gpu.py tests/test_host_logic_cpu.py -v -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
grep -n "passed\|failed\|FAILED\|Error\|rc=" $OUT/tests.log | head -20
for s in 1.0 4.0; do timeout 200 python tools/sca_frame_time.py --offsets $s >> $OUT/sca_frame_time.jsonl 2>> $OUT/sca_frame_time.err; done
cat $OUT/sca_frame_time.jsonl; tail -5 $OUT/sca_frame_time.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o sca -- python $GRAFT_REPO_ROOT/tools/sca_frame_time.py --once 10 > $OUT/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "hm5\|reduce\|plan" $f | cut -c1-220
