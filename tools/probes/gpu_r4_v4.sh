#!/bin/bash
# r04 visit 4: failing tests with tracebacks, level-class split probe, 64- vs 128-row tiles (per layer and per frame)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v4; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_bevdet_gpu.py tests/test_camera_shard_gpu.py tests/test_tile_gemm_gpu.py tests/test_int8_chain_gpu.py -q 2>&1 | tail -120 ) > $OUT/pytest_subset.log
( timeout 200 python tools/sca_split_probe.py uniform,rig 3 2>&1 | grep "^{" ) > $OUT/sca_split_probe.jsonl
( timeout 300 python tools/tile_rows_ab.py 2>&1 | grep "^{" ) > $OUT/tile_rows_ab.jsonl
( for r in 128 64 0; do
    timeout 200 python tools/model_bench.py base --graph --frames 14 --tile-rows $r 2>&1 | grep "^{" | sed "s/^{/{\"tile_rows\": $r, /"
    timeout 300 python tools/model_bench.py base --graph --int8 --frames 14 --tile-rows $r 2>&1 | grep "^{" | sed "s/^{/{\"tile_rows\": $r, /"
  done ) > $OUT/model_bench_tile_rows.jsonl
cat $OUT/pytest_subset.log | tail -90; cat $OUT/sca_split_probe.jsonl | cut -c1-260; cat $OUT/tile_rows_ab.jsonl; cat $OUT/model_bench_tile_rows.jsonl
