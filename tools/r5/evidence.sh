#!/bin/bash
# r05 evidence visit (final code): full GPU suite, smoke, the default bench line, fp16 + INT8 frame kernel traces,
# rocprofv3 kernel stats + FETCH / WRITE PMC passes of the hot-path command
TAG=${1:-r5ev}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > $OUT/pytest_gpu_tail.log
( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $OUT/smoke.log
( timeout 1200 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 ) > $OUT/bench_n1.json
bash tools/model_profile.sh $TAG/model base > $OUT/model_frame_kernel_trace.txt 2>&1; rm -rf $OUT/model
bash tools/model_profile.sh $TAG/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1; rm -rf $OUT/model_int8
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
( timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log
# the in-frame SCA sampling call alone (planned kernel + camera reduce on the rig geometry): kernel durations and the
# FETCH / WRITE bytes bench.py's roofline_frame quotes as `traffic`
P="python $GRAFT_REPO_ROOT/tools/sca_frame_time.py --once 6 --ks 2 --only planned_k2"
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sca_prof -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_prof.log
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sca_fetch -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_fetch.log
( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/sca_write -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_write.log
cd $GRAFT_REPO_ROOT
python tools/pmc_fetch_write.py "gpurun_out/$TAG (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/sca_frame_time.py --once 6 --ks 2 --only planned_k2; per-kernel means; KiB as the counters report them, FETCH not yet doubled)" $OUT/sca_fetch $OUT/sca_write msda_hm5_kernel sca_camera_reduce_kernel > $OUT/sca_plan_pmc_fetch_write.json
( grep -hE "msda_hm5_kernel|sca_camera_reduce|tsgemm" $(find $OUT/sca_prof -name "*kernel_stats.csv") | cut -c1-260 ) > $OUT/sca_plan_kernel_stats.txt
timeout 200 python tools/stem_time.py 2>> $OUT/bench.err > $OUT/stem_time.jsonl
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
du -sh $OUT; cat $OUT/sca_plan_pmc_fetch_write.json | head -20; cat $OUT/sca_plan_kernel_stats.txt; cat $OUT/pytest_gpu_tail.log | tail -6; tail -2 $OUT/smoke.log; cat $OUT/bench_n1.json | cut -c1-600; tail -3 $OUT/bench.err; head -14 $OUT/model_frame_kernel_trace.txt | cut -c1-130
