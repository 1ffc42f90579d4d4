#!/usr/bin/env python3
"""Which station bounds each kernel of the frame?  Reads the rocprofv3 --pmc --kernel-trace csv directories of a few eager
frames (tools/frame_pmc.sh) and prints, per kernel name: launches, mean duration, and per launch the counters turned
into cycles per CU -- L1 tag look-ups (TCP_TOTAL_CACHE_ACCESSES / CUs: one 64-byte sector per cycle), VALU issue
(SQ_INSTS_VALU x 4 / SIMDs), matrix-core busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMDs... as reported), LDS
(SQ_LDS_IDX_ACTIVE / CUs, of which bank conflicts), L2 requests (bytes) -- next to the duration in cycles at 2.1 GHz.
usage: frame_pmc.py dir [dir ...] [--top N]"""
import collections
import csv
import glob
import re
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 24
# launches of the run's in-process tuning (dense_auto / bevops_linear_tune candidates, MIOpen find) are not the frame's
SKIP = re.compile(r"linear_check|naive_conv|_MT(16|32|48|64|80|96)x|Histogram|MT\d+x(16|32)x\d+_|kernel_grouped_conv")
CUS, GHZ = 256, 2.1


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    return re.sub(r"^void ", "", n)[:64]


dur = collections.defaultdict(list)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (n, _), cs in per.items():
            for c, v in cs.items():
                cnt[n][c].append(v)
tot = {n: sum(v) / len(dirs) for n, v in dur.items()}
names = [n for n in sorted(tot, key=lambda n: -tot[n]) if not SKIP.search(n)][:top]
print(f"{'kernel':64s} {'n':>5s} {'us':>7s} {'kcyc':>6s} | {'L1tag':>6s} {'VALU':>6s} {'MFMA':>6s} {'LDS':>6s} {'(cnfl)':>6s} | {'L2 MB':>6s} {'L2lat':>5s}")
for n in names:
    c = {k: sum(v) / len(v) for k, v in cnt.get(n, {}).items()}
    us = sum(dur[n]) / len(dur[n])
    g = lambda k: c.get(k, float("nan"))
    l2req = g("TCP_TCC_READ_REQ_sum")
    print(f"{n:64s} {len(dur[n]) // len(dirs):5d} {us:7.1f} {us * GHZ:6.1f} | {g('TCP_TOTAL_CACHE_ACCESSES_sum') / CUS / 1e3:6.1f} "
          f"{g('SQ_INSTS_VALU') * 4 / (CUS * 4) / 1e3:6.1f} {g('SQ_VALU_MFMA_BUSY_CYCLES') / (CUS * 4) / 1e3:6.1f} "
          f"{g('SQ_LDS_IDX_ACTIVE') / CUS / 1e3:6.1f} {g('SQ_LDS_BANK_CONFLICT') / CUS / 1e3:6.1f} | {l2req * 128 / 1e6:6.1f} "
          f"{g('TCP_TCC_READ_REQ_LATENCY_sum') / max(l2req, 1):5.0f}")
