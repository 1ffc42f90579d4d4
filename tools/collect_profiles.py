#!/usr/bin/env python3
"""Copy the judged evidence of one tools/gpu_round3.sh visit into profiles/<round>/.
usage: tools/collect_profiles.py gpurun_out/<tag> profiles/<round>"""
import collections, csv, glob, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0].strip()


stats = glob.glob(os.path.join(src, "prof", "**", "bench_kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, "rocprofv3_bench_kernel_stats.csv"))
per = collections.defaultdict(dict)
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = f'{short(r["Kernel_Name"])} grid={r.get("Grid_Size", "?")}'
            vals[key].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        per[k][f"{counter}_KiB_avg"] = round(sum(v) / len(v), 1)
        per[k][f"dispatches_{counter}"] = len(v)
if per:
    json.dump(per, open(os.path.join(dst, "rocprofv3_pmc_fetch_write_per_kernel.json"), "w"), indent=1)
for name, out in (("bench.json", "bench_n1.json"), ("bench_n1.json", "bench_n1.json"), ("pytest_gpu_tail.log", "pytest_gpu_tail.log"),
                  ("shard_amdahl.jsonl", "shard_amdahl.jsonl"), ("sca_split_probe.jsonl", "sca_split_probe.jsonl"),
                  ("int8_attribution.jsonl", "int8_attribution.jsonl"), ("tile_rows_ab.jsonl", "tile_rows_ab.jsonl"), ("sweep.jsonl", "msda_sweep.jsonl"),
                  ("model_bench.jsonl", "model_bench.jsonl"), ("dcn_time.jsonl", "dcn_time.jsonl"),
                  ("pytest.log", "pytest_gpu_tail.log"), ("smoke.log", "smoke.log"),
                  ("ops_timing.jsonl", "ops_timing.jsonl"), ("bevdet_slice.jsonl", "bevdet_slice.jsonl"),
                  ("hm4_probe.jsonl", "hm4_probe.jsonl"), ("rocminfo.txt", "rocminfo.txt"),
                  ("hm5_probe.jsonl", "hm5_probe.jsonl"), ("tsgemm_time.jsonl", "tsgemm_time.jsonl"),
                  ("sca_projected_time.jsonl", "sca_projected_time.jsonl"),
                  ("model_bench_r3_ab.jsonl", "model_bench_r3_ab.jsonl"),
                  ("model_frame_kernel_trace.txt", "model_frame_kernel_trace.txt"),
                  ("model_frame_int8_kernel_trace.txt", "model_frame_int8_kernel_trace.txt"),
                  ("dense_time.jsonl", "dense_time.jsonl"), ("linear_q_time.jsonl", "linear_q_time.jsonl"),
                  ("conv_time.jsonl", "conv_time.jsonl"), ("int8_model_delta.jsonl", "int8_model_delta.jsonl"),
                  ("model_bench_small_ab.jsonl", "model_bench_small_ab.jsonl"),
                  ("sca_plan_pmc_fetch_write.json", "sca_plan_pmc_fetch_write.json"),
                  ("sca_plan_kernel_stats.txt", "sca_plan_kernel_stats.txt"), ("stem_time.jsonl", "stem_time.jsonl")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        lines = [l for l in open(p) if "amdgpu.ids" not in l]
        open(os.path.join(dst, out), "w").writelines(lines)
print(sorted(os.listdir(dst)))
