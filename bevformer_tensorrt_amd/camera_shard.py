"""Camera sharding of the SCA sampler across GPUs (new design --
the reference is single-GPU).  One process per GPU; rank g owns cameras
{c : c mod G == g}; after the per-camera MSDA each rank holds [cams_local, nq, embed] and the
encoder layer's exchange (RCCL over xGMI, backend "nccl") rebuilds what the replicated masked camera
sum + output_proj need (det2trt/models/modules/spatial_cross_attention.py:270-273).  Two exchanges,
both behind `CameraExchange`:

  "gather"  one asynchronous all_gather_into_tensor PER LOCAL CAMERA SLOT (20.5 MB at base fp16),
            issued as soon as that camera has been sampled, so camera i's transfer overlaps the
            sampling of camera i + 1; waited for right before the masked sum;
  "reduce"  each rank reduces its own cameras with the bev_mask weights and ONE all-reduce per layer
            adds the [1, nq, embed] partial sums (6x less data).

  "scatter" (round 5) the cameras are sharded as above AND the rest of the encoder by QUERY RANGE (SURVEY.md 8e:
            "replicate, or shard by query range -- MSDA is also independent per query"): rank g keeps rows
            [g * per, (g + 1) * per) of the BEV queries, per = ceil(nq / G).  Temporal self-attention, the three
            LayerNorms, the FFN and both output projections run on the local rows only; per layer ONE all-gather
            rebuilds the full query tensor in front of spatial cross-attention (its sampling offsets / weights are
            needed for every query by every camera's sampler) and ONE reduce-scatter -- instead of the all-reduce --
            hands every rank the masked camera sum of its own rows.  Same bytes on the wire as "reduce" (an all-reduce
            IS a reduce-scatter plus an all-gather), but the replicated per-query work (3.2 ms of a 12.5 ms base frame)
            is divided by G.

`gather_camera_features` is the plain one-collective form (used by bench.py's hot-path step and the
primitive tests).  Staging buffers are keyed by device AND current stream: two streams (or a graph
capture next to eager work) never share one.
"""
import torch

_BUFFERS = {}


def camera_shards(n_cams, world):
    """List (indexed by rank) of the camera ids each rank owns."""
    return [[c for c in range(n_cams) if c % world == r] for r in range(world)]


def _buffers(key, max_local, world, tail, dtype, device):
    hit = _BUFFERS.get(key)
    if hit is None:
        send = torch.zeros((max_local,) + tail, dtype=dtype, device=device)
        recv = torch.empty((world, max_local) + tail, dtype=dtype, device=device)
        hit = (send, recv)
        _BUFFERS[key] = hit
    return hit


def gather_camera_features(local, n_cams, dist, group=None):
    """all-gather per-camera features.

    local: [cams_local, ...] for this rank's cameras (camera_shards order).
    Returns [n_cams, ...] in global camera order on every rank.  Uneven shards
    (e.g. 6 cameras on 4 or 8 ranks) are padded to ceil(n_cams / world) slots.
    """
    world = dist.get_world_size(group)
    max_local = -(-n_cams // world)
    tail = tuple(local.shape[1:])
    stream = torch.cuda.current_stream(local.device).cuda_stream if local.is_cuda else 0
    key = (max_local, world, tail, local.dtype, str(local.device), stream)
    send, recv = _buffers(key, max_local, world, tail, local.dtype, local.device)
    if local.shape[0] == max_local and local.is_contiguous():
        src = local
    else:
        send[: local.shape[0]].copy_(local)
        src = send
    dist.all_gather_into_tensor(recv.view(-1), src.view(-1), group=group)
    # recv[r, i] is camera r + world*i  ->  camera-major order
    return recv.transpose(0, 1).reshape((max_local * world,) + tail)[:n_cams]


def reduce_camera_slots(local_weighted_sum, dist, group=None):
    """The cheaper exchange of SURVEY.md 8e ("note for the builder"): every rank first reduces ITS
    cameras -- sum over local cameras of bev_mask * sampled, [1, nq, embed] = 20.5 MB at base fp16
    -- and ONE all-reduce per encoder layer adds the partial sums (6x less data than gathering the
    per-camera features; bev_mask is known on every rank because it only depends on lidar2img).
    In place on `local_weighted_sum`; ranks without cameras pass zeros."""
    dist.all_reduce(local_weighted_sum, op=dist.ReduceOp.SUM, group=group)
    return local_weighted_sum


class CameraExchange:
    """The per-layer exchange of the camera-sharded encoder, as one object the model calls
    (`BEVFormer.forward(..., cams=..., gather=CameraExchange(...))`).

    mode "gather" (BASELINE config 4): the local cameras are sampled ONE AT A TIME and each
    camera's [nq, embed] features go out as their own asynchronous all-gather -- the collective
    of camera i runs (on the process group's stream) while camera i + 1 is being sampled, so only
    the last camera's exchange is exposed.  Camera c = i * world + rank lands in slot (i, rank) of
    the receive buffer, which therefore already IS camera-major.  Ranks with fewer cameras than
    ceil(n_cams / world) still join every collective (with zeros).
    mode "reduce" (SURVEY.md 8e alternative): each rank reduces its cameras with the bev_mask
    weights and ONE all-reduce adds the [1, nq, embed] partial sums."""

    def __init__(self, dist, n_cams, mode="gather", group=None):
        assert mode in ("gather", "reduce", "scatter")
        self.dist, self.n_cams, self.mode, self.group = dist, n_cams, mode, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.cams = camera_shards(n_cams, self.world)[self.rank]
        self.max_local = -(-n_cams // self.world)
        self._recv = {}

    def gather(self, sample, tail, dtype, device):
        """sample(i) -> [1, *tail] features of the i-th LOCAL camera.  Returns [n_cams, *tail]."""
        dev = torch.device(device)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (tuple(tail), dtype, str(device), stream)
        recv = self._recv.get(key)
        if recv is None:
            recv = self._recv[key] = torch.empty((self.max_local, self.world) + tuple(tail), dtype=dtype, device=device)
        works, keep = [], []
        for i in range(self.max_local):
            if i < len(self.cams):
                out = sample(i).reshape(-1)
                if not out.is_contiguous():
                    out = out.contiguous()
            else:
                out = torch.zeros(recv[i, 0].numel(), dtype=dtype, device=device)
            keep.append(out)   # alive until its collective has run
            works.append(self.dist.all_gather_into_tensor(recv[i].view(-1), out, group=self.group, async_op=True))
        for w in works:
            w.wait()
        return recv.view((self.max_local * self.world,) + tuple(tail))[: self.n_cams]

    def reduce(self, local_weighted_sum):
        return reduce_camera_slots(local_weighted_sum, self.dist, self.group)

    # ---- query-range sharding (mode "scatter")
    def query_range(self, nq):
        """(lo, hi, per): this rank's rows of the [nq, .] query tensor and the padded rows per rank."""
        per = -(-nq // self.world)
        lo = min(self.rank * per, nq)
        return lo, min(lo + per, nq), per

    def _q_buffers(self, nq, width, dtype, device):
        dev = torch.device(device)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = ("q", nq, width, dtype, str(device), stream)
        hit = self._recv.get(key)
        if hit is None:
            per = -(-nq // self.world)
            hit = self._recv[key] = (torch.zeros((self.world * per, width), dtype=dtype, device=device),
                                     torch.zeros((per, width), dtype=dtype, device=device))
        return hit

    @torch.no_grad()
    def all_gather_queries(self, local, nq):
        """local [1, rows of this rank, width] -> the full [1, nq, width] on every rank (one all-gather).  Inference
        only (as the whole exchange object): no gradient flows through the collectives."""
        local = local.detach()
        width = local.shape[-1]
        lo, hi, per = self.query_range(nq)
        full, mine = self._q_buffers(nq, width, local.dtype, local.device)
        if hi - lo == per and local.is_contiguous():
            src = local.view(per, width)
        else:                                   # the last rank's range may be short (or empty): pad with zeros
            mine[: hi - lo].copy_(local.view(hi - lo, width))
            src = mine
        self.dist.all_gather_into_tensor(full.view(-1), src.view(-1), group=self.group)
        return full[:nq].view(1, nq, width).clone()

    @torch.no_grad()
    def reduce_scatter_queries(self, partial, nq):
        """partial [1, nq, width] (this rank's masked camera sum over all queries) -> the sum over ranks of the rows
        this rank owns, [1, hi - lo, width] (one reduce-scatter)."""
        partial = partial.detach()
        width = partial.shape[-1]
        lo, hi, per = self.query_range(nq)
        full, mine = self._q_buffers(nq, width, partial.dtype, partial.device)
        if nq == self.world * per and partial.is_contiguous():
            src = partial.view(nq, width)
        else:
            full[:nq].copy_(partial.view(nq, width))
            full[nq:].zero_()
            src = full
        out = torch.empty((per, width), dtype=partial.dtype, device=partial.device)
        self.dist.reduce_scatter_tensor(out.view(-1), src.view(-1), op=self.dist.ReduceOp.SUM, group=self.group)
        return out[: hi - lo].view(1, hi - lo, width)
