#!/bin/bash
# final r02 evidence: full suite, smoke, default bench line, rocprof stats + PMC passes of the same command,
# torchrun-launched one-rank bench (the driver's launch form), op timings
TAG=r02z
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 ) > $OUT/pytest.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $OUT/smoke.log
( timeout 900 python bench.py 2>&1 | tail -1 ) > $OUT/bench.json
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-geometry-extra 2>&1 | grep "^{" | tail -1 ) > $OUT/bench_torchrun_n1.json
( timeout 300 python tools/ops_timing.py 2>&1 | grep "{" ) > $OUT/ops_timing.jsonl
( timeout 300 python tools/dcn_int8_time.py 2>&1 | grep "{" ) > $OUT/dcn_int8_time.jsonl
( timeout 600 python tools/model_bench.py --graph 2>&1 | grep "{" ) > $OUT/model_bench.jsonl
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- $B 2>&1 | tail -3 ) > $OUT/rocprof.log
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-geometry-extra"
( timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_fetch.log
( timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B2 2>&1 | tail -2 ) > $OUT/rocprof_pmc_write.log
cd $GRAFT_REPO_ROOT
find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT; tail -4 $OUT/pytest.log; tail -2 $OUT/smoke.log; cat $OUT/bench.json; echo; cat $OUT/bench_torchrun_n1.json | cut -c1-400; cat $OUT/model_bench.jsonl
