#!/bin/bash
# r05 visit 4: operand-row warming A/B of the planned SCA kernel; first bench line with roofline_frame / roofline_mfma / tiny
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5v4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/sca_frame_time.py --ks 2,3,4 --warm > $OUT/sca_warm_ab.jsonl 2> $OUT/err.log; cat $OUT/sca_warm_ab.jsonl; tail -3 $OUT/err.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_n1.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline_frame", "roofline_mfma", "tiny", "dispatch_misses"):
    print(k, json.dumps(d.get(k))[:600])
print("int8", json.dumps(d["int8"].get("end_to_end"))[:300])
print("roofline", json.dumps({k: v for k, v in d["roofline"].items() if k in ("frac", "avg_launch_us")}))
print("cpu", json.dumps(d["cpu_baseline"])[:900])
PY
