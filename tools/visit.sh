#!/bin/bash
# One GPU visit = one call of this script under gpurun:  gpurun --timeout S -- 'bash tools/visit.sh TAG STEP [STEP ...]'
# Every step writes under gpurun_out/TAG/ (merged back by gpurun); the summaries worth judging are copied to
# profiles/<round>/ afterwards (tools/collect_profiles.py).  Steps (each bounded by its own timeout):
#   tests[:EXPR]   GPU suite (pytest -m gpu [-k EXPR]) -> pytest_gpu_tail.log
#   testsv:EXPR    the same with -s (the tests' printed error figures) -> pytest_gpu_verbose.log
#   smoke          __graft_entry__.smoke()              -> smoke.log
#   bench[:ARGS]   python bench.py ARGS (default --steps 20 --warmup 5) -> bench_n1.json (+ bench.err)
#   trace[:int8]   kernel trace of one base frame (tools/model_profile.sh) -> model_frame[_int8]_kernel_trace.txt
#   prof           rocprofv3 --kernel-trace --stats of the hot-path bench command -> prof/
#   pmc            FETCH_SIZE / WRITE_SIZE passes of the same command (separate runs) -> pmc_fetch/, pmc_write/
#   sca            the in-frame SCA sampling call: kernel stats + FETCH / WRITE passes -> sca_plan_*.{txt,json}
#   scapmc[:NAMES] SQ / LDS / TCP counters + kernel stats of flavours of tools/sca_frame_time.py -> sca_pmc.txt
#   kpmc:SUBSTR:SCRIPT ARGS   SQ / LDS / TCP counters + kernel stats of the kernels named *SUBSTR* in python tools/SCRIPT ARGS -> kernel_pmc.txt
#   framepmc       SQ / TCP / TCC counters of every kernel of the frame (tools/frame_pmc.sh)
#   py:SCRIPT ARGS python tools/SCRIPT ARGS (quote the step) -> SCRIPT.jsonl
TAG=${1:?tag}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > $OUT/rocminfo.txt 2>&1; nproc >> $OUT/rocminfo.txt
HOT="python $GRAFT_REPO_ROOT/bench.py --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$name" != "$step" ] && arg=${step#*:}
  cd $GRAFT_REPO_ROOT
  case $name in
    tests) ( timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider ${arg:+-k "$arg"} 2>&1 | tail -40 ) > $OUT/pytest_gpu_tail.log; tail -8 $OUT/pytest_gpu_tail.log ;;
    testsv) ( timeout 1700 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "$arg" 2>&1 | grep -vE "^\s*$" | tail -200 ) > $OUT/pytest_gpu_verbose.log; grep -E "passed|failed|err|agreement|max " $OUT/pytest_gpu_verbose.log | tail -30 ;;
    smoke) ( timeout 200 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $OUT/smoke.log; tail -2 $OUT/smoke.log ;;
    bench) ( timeout 1500 python bench.py ${arg:---steps 20 --warmup 5} 2>$OUT/bench.err | tail -1 ) > $OUT/bench_n1.json; cut -c1-700 $OUT/bench_n1.json; tail -3 $OUT/bench.err ;;
    trace) if [ "$arg" = int8 ]; then bash tools/model_profile.sh $TAG/model_int8 base --int8 > $OUT/model_frame_int8_kernel_trace.txt 2>&1; rm -rf $OUT/model_int8
           else bash tools/model_profile.sh $TAG/model base > $OUT/model_frame_kernel_trace.txt 2>&1; rm -rf $OUT/model; head -16 $OUT/model_frame_kernel_trace.txt | cut -c1-130; fi ;;
    prof) cd /tmp; ( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $HOT --steps 5 2>&1 | tail -3 ) > $OUT/rocprof.log ;;
    pmc) cd /tmp
         ( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $HOT --steps 2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
         ( timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $HOT --steps 2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log ;;
    sca) cd /tmp; P="python $GRAFT_REPO_ROOT/tools/sca_frame_time.py --once 6 --ks 2 --only planned_k2"
         ( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sca_prof -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_prof.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/sca_fetch -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_fetch.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/sca_write -o p -- $P 2>&1 | tail -2 ) > $OUT/sca_write.log
         cd $GRAFT_REPO_ROOT
         python tools/pmc_fetch_write.py "gpurun_out/$TAG (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/sca_frame_time.py --once 6 --ks 2 --only planned_k2; per-kernel means; KiB as the counters report them, FETCH not yet doubled)" $OUT/sca_fetch $OUT/sca_write msda_hm5 sca_camera_reduce_kernel > $OUT/sca_plan_pmc_fetch_write.json
         ( grep -hE "msda_hm5|sca_camera_reduce|tsgemm" $(find $OUT/sca_prof -name "*kernel_stats.csv") | cut -c1-260 ) > $OUT/sca_plan_kernel_stats.txt; cat $OUT/sca_plan_kernel_stats.txt ;;
    scapmc) cd /tmp; P="python $GRAFT_REPO_ROOT/tools/sca_frame_time.py --once 6 --ks 2 --only ${arg:-planned_k2}"
         ( timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/scapmc1 -o p -- $P 2>&1 | tail -2 ) > $OUT/scapmc1.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/scapmc2 -o p -- $P 2>&1 | tail -2 ) > $OUT/scapmc2.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/scapmc3 -o p -- $P 2>&1 | tail -2 ) > $OUT/scapmc3.log
         ( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/scapmc0 -o p -- $P 2>&1 | tail -2 ) > $OUT/scapmc0.log
         cd $GRAFT_REPO_ROOT; python tools/pmc_table.py msda_hm5_kernel $OUT/scapmc1 $OUT/scapmc2 $OUT/scapmc3 > $OUT/sca_pmc.txt 2>&1
         ( grep -hE "msda_hm5|sca_camera_reduce" $(find $OUT/scapmc0 -name "*kernel_stats.csv") | cut -c1-200 ) >> $OUT/sca_pmc.txt; cat $OUT/sca_pmc.txt ;;
    kpmc) sub=${arg%%:*}; cmd=${arg#*:}; cd /tmp; P="python $GRAFT_REPO_ROOT/tools/$cmd"
         ( timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/kpmc1 -o p -- $P 2>&1 | tail -2 ) > $OUT/kpmc1.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $OUT/kpmc2 -o p -- $P 2>&1 | tail -2 ) > $OUT/kpmc2.log
         ( timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/kpmc3 -o p -- $P 2>&1 | tail -2 ) > $OUT/kpmc3.log
         ( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kpmc0 -o p -- $P 2>&1 | tail -2 ) > $OUT/kpmc0.log
         cd $GRAFT_REPO_ROOT; python tools/pmc_table.py "$sub" $OUT/kpmc1 $OUT/kpmc2 $OUT/kpmc3 > $OUT/kernel_pmc.txt 2>&1
         ( grep -hE "$sub" $(find $OUT/kpmc0 -name "*kernel_stats.csv") | cut -c1-60,300-420 ) >> $OUT/kernel_pmc.txt; cat $OUT/kernel_pmc.txt ;;
    framepmc) KINDS=${arg:-fp16} bash tools/frame_pmc.sh $TAG ;;
    py) set -- $arg; s=$1; shift; ( timeout 900 python tools/$s "$@" 2>>$OUT/py.err ) > $OUT/$(basename $s .py).jsonl; tail -20 $OUT/$(basename $s .py).jsonl | cut -c1-400; tail -3 $OUT/py.err ;;
    *) echo "unknown step $step" ;;
  esac
done
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
du -sh $OUT
