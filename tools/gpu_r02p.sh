#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_mdconv_gpu.py tests/test_quantization_gpu.py tests/test_int8_model_gpu.py -q -x 2>&1 | tail -5 ) > $OUT/pytest.log
( timeout 300 python tools/dcn_int8_time.py 2>&1 | grep "{" ) > $OUT/dcn_int8_time.jsonl
( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | tail -1 ) > $OUT/bench.json
tail -3 $OUT/pytest.log; grep dcn_ $OUT/dcn_int8_time.jsonl
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['int8']['value'], d['int8']['ms_per_step'], d['int8']['roofline']['avg_launch_us'], d['roofline']['avg_launch_us'])"
