#!/bin/bash
# r05 visit 3: what a round of the planned SCA kernel costs on the rig geometry (ablations, SQ / TCP counters)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/sca_frame_time.py --ks 2 --ablate > $OUT/sca_ablate.jsonl 2> $OUT/err.log; cat $OUT/sca_ablate.jsonl; tail -3 $OUT/err.log
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/sca_frame_time.py --once 4 --ks 2"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -o p -- $P > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- $P > $OUT/pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -o p -- $P > $OUT/pmc3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc4 -o p -- $P > $OUT/pmc4.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc5 -o p -- $P > $OUT/pmc5.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py hm5_kernel $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 | tee $OUT/pmc_table.txt
tail -2 $OUT/pmc1.log | cut -c1-200
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
