#!/bin/bash
# counters of every kernel of the fp16 (and INT8) base frame: three passes over a few eager frames
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}/framepmc; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for kind in ${KINDS:-fp16 int8}; do
  EXTRA=""; [ $kind = int8 ] && EXTRA="--int8"
  P="python $GRAFT_REPO_ROOT/tools/model_bench.py base --frames 3 $EXTRA"
  timeout 400 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$kind/p1 -o p -- $P > $OUT/${kind}_p1.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/$kind/p2 -o p -- $P > $OUT/${kind}_p2.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$kind/p3 -o p -- $P > $OUT/${kind}_p3.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/frame_pmc.py $OUT/$kind/p1 $OUT/$kind/p2 $OUT/$kind/p3 --top 30 > $OUT/frame_pmc_$kind.txt 2>&1
  cat $OUT/frame_pmc_$kind.txt | cut -c1-200
  cd /tmp
done
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*.csv" -size +1M -delete
