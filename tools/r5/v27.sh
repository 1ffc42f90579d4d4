#!/bin/bash
# r05 visit 27: counters of the one-kernel stem at the base shape
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v27; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/stem_time.py --once 3"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -o p -- $P > $OUT/pmc1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- $P > $OUT/pmc2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $P > $OUT/pmc3.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o p -- $P > $OUT/pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py stem7x7 $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 | tee $OUT/stem_pmc.txt
tail -2 $OUT/pmc2.log | cut -c1-200
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +1M -delete
