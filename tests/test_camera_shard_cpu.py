"""world_size-2/4 gloo tests (CPU) of the camera-sharded exchange used for N>1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_cams, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bevformer_tensorrt_amd.camera_shard import camera_shards, gather_camera_features
    full = torch.arange(n_cams * 5 * 3, dtype=torch.float32).view(n_cams, 5, 3)
    mine = camera_shards(n_cams, world)[rank]
    local = full[mine] if mine else full[:0]
    for _ in range(2):  # second call reuses cached buffers
        got = gather_camera_features(local, n_cams, dist)
    ok = bool(torch.equal(got, full))
    # the all-reduce alternative: every rank reduces its cameras with the mask weights, the ranks'
    # partial sums add up to the masked camera sum of the gathered features
    from bevformer_tensorrt_amd.camera_shard import reduce_camera_slots
    mask = (torch.arange(n_cams * 5, dtype=torch.float32).view(n_cams, 5, 1) % 3) / 2.0
    want = (full * mask).sum(0, keepdim=True)
    part = (local * mask[mine]).sum(0, keepdim=True) if mine else torch.zeros(1, 5, 3)
    ok = ok and bool(torch.allclose(reduce_camera_slots(part, dist), want))
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cams", [(2, 6), (4, 6), (2, 5), (8, 6)])   # 8: two ranks own no camera
def test_gather_camera_features_gloo(world, n_cams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_camera_shards_partition():
    from bevformer_tensorrt_amd.camera_shard import camera_shards
    for world in (1, 2, 4, 8):
        sh = camera_shards(6, world)
        assert len(sh) == world
        assert sorted(c for s in sh for c in s) == list(range(6))
        assert all(c % world == r for r, s in enumerate(sh) for c in s)
