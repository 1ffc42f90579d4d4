#!/bin/bash
# r02 visit C: hm4 int8 parity after the un-fused softmax, ablation timings, PMC of the hm4 kernels
OUT=gpurun_out/r02c; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_msda_hm4_gpu.py tests/test_msda_int8_gpu.py -q 2>&1 | tail -40 ) > $OUT/pytest_hm4.log
( timeout 600 python tools/hm4_probe.py ablate 2>&1 | grep "{" ) > $OUT/hm4_ablate.jsonl
( rocprofv3 -L 2>&1 | grep -E "^\s*(Name|gfx|SQ_|TCP_|TA_|TCC_|GRBM_|TD_)" | head -400 ) > $OUT/counters.txt
( bash tools/pmc_probe.sh r02c/pmc_f16 base_sca 17 2>&1 | tail -60 ) > $OUT/pmc_f16.txt
tail -12 $OUT/pytest_hm4.log; cat $OUT/hm4_ablate.jsonl; cat $OUT/pmc_f16.txt; wc -l $OUT/counters.txt
