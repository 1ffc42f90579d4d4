#!/bin/bash
# r05 visit 10: offset convolution with quad-coalesced image loads + register transpose: tests, timing, counters
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v10; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_mdconv_gpu.py -q -p no:cacheprovider -x -k "conv_offset or nhwc or packed" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
timeout 300 python tools/conv_offset_time.py > $OUT/conv_offset_time.jsonl 2> $OUT/err.log; cat $OUT/conv_offset_time.jsonl; tail -3 $OUT/err.log
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/conv_offset_time.py --once"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc1 -o p -- $P > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc2 -o p -- $P > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py "conv3x3_c32_kernel<4, 5" $OUT/pmc1 $OUT/pmc2 | tee $OUT/pmc_table.txt
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
timeout 600 python tools/model_bench.py base --graph --frames 30 > $OUT/model_bench.jsonl 2>> $OUT/err.log; cat $OUT/model_bench.jsonl
