#!/bin/bash
# r04 visit 12: int8 SCA big set as pixel-pair entries (variants 21 / 22) against 2x2 footprints (23 / 24): bits, time, traffic
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v12; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_msda_hm4_gpu.py -q -k "int8 or prepacked" 2>&1 | tail -8 ) > $OUT/pytest_hm4_int8.log
( timeout 200 python tools/probes/msda_i8_ab.py 23 21 3; timeout 200 python tools/probes/msda_i8_ab.py 24 22 3 ) > $OUT/msda_i8_pair_ab.jsonl 2>$OUT/ab.err
cd /tmp
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o i8 -- python $GRAFT_REPO_ROOT/tools/probes/msda_i8_traffic.py 23 21 24 22 2>&1 | tail -2 ) > $OUT/pmc.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete
cat $OUT/pytest_hm4_int8.log; cat $OUT/msda_i8_pair_ab.jsonl; tail -3 $OUT/ab.err; cat $OUT/pmc.log
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4v12/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hm4" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:70], r["Counter_Name"], r["Counter_Value"])
PY
