#!/bin/bash
# Kernel-level breakdown of one steady-state frame of the re-hosted model (eager, so every kernel
# has its own name).  usage: tools/model_profile.sh <tag> [model] [--int8]
TAG=$1; MODEL=${2:-base}; EXTRA=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o m -- python $GRAFT_REPO_ROOT/tools/model_bench.py $MODEL --frames 6 $EXTRA > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# frames are delimited by the rotate kernel (one per frame after the first)
idx = [i for i, r in enumerate(rows) if "rotate_kernel" in r["Kernel_Name"] or "rotate_hwc_kernel" in r["Kernel_Name"]]
a, b = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows))
fr = rows[a:b]
span = (int(fr[-1]["End_Timestamp"]) - int(fr[0]["Start_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in fr:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"frame span {span:.2f} ms, kernels {len(fr)}, busy {tot/1e3:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{t/1e3:8.3f} ms {100*t/tot:5.1f}%  x{c:<4d} {n}")
PY
