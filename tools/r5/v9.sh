#!/bin/bash
# r05 visit 9: counters of the offset convolution
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v9; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
P="python $GRAFT_REPO_ROOT/tools/conv_offset_time.py --once"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc1 -o p -- $P > $OUT/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc2 -o p -- $P > $OUT/pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o p -- $P > $OUT/pmc3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -o p -- $P > $OUT/pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_table.py conv3x3_c32 $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 | tee $OUT/pmc_table.txt
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3_c32" in r["Kernel_Name"]:
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            agg[(n, r.get("Grid_Size"), r.get("Workgroup_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(k, "x%d avg %.2f us" % (len(v), sum(v) / len(v)))
PY
tail -2 $OUT/pmc4.log | cut -c1-160
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
