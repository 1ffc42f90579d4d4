"""bench.py's multi-GPU code path on a 1-GPU box: `--sharded-path` makes rank 0 of a ONE-rank RCCL group run exactly
what every rank of `--gpus N` runs -- process group, CameraExchange, camera + query-range sharding, the frame's HIP
graph with its collectives inside, barrier + max-over-ranks timing -- and print the contract line.  fp16 with the
default "scatter" exchange, and the INT8 engine (which bench.py refused at N > 1 until round 5)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-path", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-hot-path", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]            # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,port", [((), 29631), (("--exchange", "reduce"), 29632), (("--dtype", "int8"), 29633)])
def test_sharded_bench_line_on_one_rank(extra, port):
    d = _bench(*extra, port=port)
    assert d["ranks_seen"] == 1 and d["n_gpus"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["hip_graph"] is True                      # captured with its RCCL collectives, not an eager fallback
    want = "reduce" if "reduce" in extra else "scatter"
    assert d["config"]["parallelism"] == f"cameras/1+{want}"
    assert d["dtype"] == ("i8" if "int8" in extra else "f16")
    assert d["scaling"] == "strong" and d["unit"] == "frames/s"
