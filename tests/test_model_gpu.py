"""Model-level dataflow check: the re-hosted BEVFormer wrappers evaluated with
the HIP operators must agree with the same network (same random weights) evaluated with the
reference's PyTorch formulations of those operators, over a 3-frame sequence with a scene
reset, for the tiny config (no DCN) in fp32, and a DCN-bearing cut-down config."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def frames(cfg_image, n, dev, dtype):
    g = torch.Generator().manual_seed(0)
    H, W = cfg_image
    out = []
    for i in range(n):
        img = torch.randn(1, 6, 3, H, W, generator=g).to(dev, dtype)
        can = torch.zeros(18)
        can[0], can[1], can[-2], can[-1] = 0.4 * i, -0.15 * i, 0.02 * i, 1.1 * i
        out.append((img, can, "scene0" if i < 2 else "scene1"))
    return out


def run_sequence(name, ops, dtype, n=3, graph=False):
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    dev = torch.device("cuda")
    model = B.BEVFormer(name, ops=ops, seed=0).to(dev, dtype)
    runner = B.FrameRunner(model, dev, dtype, graph=graph)
    l2i = G.synthetic_lidar2img(B.CONFIGS[name]["image"]).to(dev)
    outs = []
    for img, can, scene in frames(B.CONFIGS[name]["image"], n, dev, dtype):
        cls, crd = runner.step(img, can, l2i, scene)
        outs.append((runner.prev_bev.float().clone(), cls.float().clone(), crd.float().clone()))
    return outs


def test_tiny_fp32_matches_reference_formulation():
    import bevformer_tensorrt_amd.functions as hip_ops
    from util_refops import RefOps
    a = run_sequence("tiny", hip_ops, torch.float32)
    b = run_sequence("tiny", RefOps, torch.float32)
    for (bev_a, cls_a, crd_a), (bev_b, cls_b, crd_b) in zip(a, b):
        assert bev_a.shape == (2500, 1, 256) and cls_a.shape == (6, 1, 900, 10) and crd_a.shape == (6, 1, 900, 10)
        assert torch.isfinite(bev_a).all()
        assert (bev_a - bev_b).abs().max().item() <= 2e-3
        assert (cls_a - cls_b).abs().max().item() <= 2e-3
        assert (crd_a - crd_b).abs().max().item() <= 5e-3


def test_tiny_fp16_runs_and_tracks_fp32():
    import bevformer_tensorrt_amd.functions as hip_ops
    a = run_sequence("tiny", hip_ops, torch.float16)
    b = run_sequence("tiny", hip_ops, torch.float32)
    for (bev_a, cls_a, _), (bev_b, cls_b, _) in zip(a, b):
        assert torch.isfinite(bev_a).all()
        assert (bev_a - bev_b).abs().mean().item() <= 5e-2   # 3-6 transformer layers in fp16


def test_dcn_backbone_block_matches_reference_formulation():
    """ResNet bottleneck with DCNv2Pack (stage-3 style) vs the oracle DCN, fp32, small map."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    from util_refops import RefOps
    torch.manual_seed(0)
    x = torch.randn(2, 256, 20, 28).cuda()
    blk_a = B.Bottleneck(256, 64, 1, True, hip_ops, True).cuda().eval()
    blk_b = B.Bottleneck(256, 64, 1, True, RefOps, True).cuda().eval()
    blk_b.load_state_dict(blk_a.state_dict())
    with torch.no_grad():
        ya, yb = blk_a(x), blk_b(x)
    assert (ya - yb).abs().max().item() <= 1e-3


def test_graph_replay_equals_eager():
    """The HIP-graph frame loop (static buffers, prev_bev updated inside the graph) must give
    what the eager loop gives, including across the scene reset."""
    import bevformer_tensorrt_amd.functions as hip_ops
    a = run_sequence("tiny", hip_ops, torch.float16, n=4, graph=False)
    b = run_sequence("tiny", hip_ops, torch.float16, n=4, graph=True)
    for (bev_a, cls_a, crd_a), (bev_b, cls_b, crd_b) in zip(a, b):
        # same kernels, but MIOpen / hipBLASLt may pick other algorithms under capture:
        # fp16 rounding noise through 3 transformer layers, not a dataflow difference
        for x, y in ((bev_a, bev_b), (cls_a, cls_b)):
            scale = max(1.0, x.abs().max().item())
            assert (x - y).abs().max().item() <= 4e-2 * scale
            assert (x - y).abs().mean().item() <= 4e-3 * scale


def test_point_sampling_fma_projection_matches_matmul():
    from bevformer_tensorrt_amd import geometry as G
    ref3d = G.reference_points_3d(50, 50, 8, 4, device="cuda")
    l2i = G.synthetic_lidar2img((480, 800)).cuda()
    pc = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    cam_a, mask_a = G.point_sampling(ref3d, pc, l2i, (480, 800))
    cam_b, mask_b = G.point_sampling(ref3d, pc, l2i, (480, 800), projection="fma")
    assert (cam_a - cam_b).abs().max().item() <= 1e-5 * max(1.0, cam_a.abs().max().item())
    assert (mask_a != mask_b).float().mean().item() <= 1e-4   # only points within 1 ulp of a frustum edge


def test_channels_last_backbone_matches_reference_layout_path():
    """ResNet(+DCNv2)+FPN: the NHWC data path (1x1 convs as GEMMs with fused shift/ReLU, NHWC
    library convs + one bias_act pass, NHWC DCNv2) against the plain NCHW path, same weights."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    torch.manual_seed(0)
    net = B.ResNet(50, (False, False, True, True), (1, 2, 3), hip_ops).cuda().half().eval()
    neck = B.FPN([512, 1024, 2048], 256, 4).cuda().half().eval()
    x = torch.randn(2, 3, 192, 256, device="cuda", dtype=torch.half)
    with torch.no_grad():
        ref = neck(net(x))
        for m in list(net.modules()) + list(neck.modules()):
            if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        got = neck.forward_nhwc(net.forward_nhwc(x, hip_ops), hip_ops)
    assert len(ref) == len(got) == 4
    for a, b in zip(ref, got):
        assert a.shape == b.shape and b.is_contiguous(memory_format=torch.channels_last)
        scale = max(1.0, a.abs().max().item())
        assert (a.float() - b.float()).abs().max().item() <= 3e-2 * scale
        assert (a.float() - b.float()).abs().mean().item() <= 3e-3 * scale


def test_bias_act_nhwc_matches_torch():
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(0)
    for shape in ((2, 64, 17, 23), (1, 256, 5, 7), (3, 24, 9, 4)):
        x = torch.randn(*shape, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
        r = torch.randn(*shape, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
        b = torch.randn(shape[1], generator=g).half().cuda()
        for res in (None, r):
            for relu in (False, True):
                want = x.float() + b.float().view(1, -1, 1, 1) + (res.float() if res is not None else 0)
                want = torch.relu(want) if relu else want
                got = bev.bias_act_nhwc_(x.clone(memory_format=torch.preserve_format), b, res, relu)
                assert (got.float() - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())


def test_linear_bias_act_matches_torch():
    """bevops_linear_bias_act (one hipBLASLt GEMM with shift + identity + ReLU in its epilogue) vs
    the op sequence it replaces, at backbone / encoder shapes incl. a ragged row count; in-place on
    the residual; graph-capturable."""
    import torch.nn.functional as F
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd.utils.lib import BevopsError
    g = torch.Generator().manual_seed(0)
    for M, K, N in [(6 * 58 * 100, 256, 1024), (40000, 512, 256), (1237, 64, 256), (900, 256, 256)]:
        x = torch.randn(M, K, generator=g).half().cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
        b = torch.randn(N, generator=g).half().cuda()
        r = torch.randn(M, N, generator=g).half().cuda()
        for bias, res, relu in [(b, r, True), (b, None, True), (b, r, False), (None, None, False), (None, r, True)]:
            want = x.float() @ w.float().t()
            if bias is not None:
                want = want + bias.float()
            if res is not None:
                want = want + res.float()
            if relu:
                want = F.relu(want)
            try:
                got = hip_ops.linear_bias_act(x, w, bias, res, relu)
            except BevopsError as exc:   # no library algorithm for this shape: the model falls back
                assert exc.status == 3
                continue
            assert got.shape == (M, N)
            err = (got.float() - want).abs()
            assert err.max().item() <= 4e-3 * max(1.0, want.abs().max().item()), (M, K, N, err.max().item())
        r2 = r.clone()
        try:
            hip_ops.linear_bias_act(x, w, b, r2, True, out=r2)          # in place on the residual
            want = F.relu(x.float() @ w.float().t() + b.float() + r.float())
            assert (r2.float() - want).abs().max().item() <= 4e-3 * max(1.0, want.abs().max().item())
        except BevopsError as exc:
            assert exc.status == 3


def test_fused_linear_model_path_equals_two_launch_path():
    """The re-hosted model with the fused GEMM epilogues vs the same weights on the two-launch path."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    B._FUSED_LINEAR["enabled"] = True
    a = run_sequence("tiny", hip_ops, torch.float16, n=2)
    B._FUSED_LINEAR["enabled"] = False
    try:
        b = run_sequence("tiny", hip_ops, torch.float16, n=2)
    finally:
        B._FUSED_LINEAR["enabled"] = True
    for fa, fb in zip(a, b):
        for x, y in zip(fa, fb):
            scale = max(1.0, y.float().abs().max().item())
            assert (x.float() - y.float()).abs().max().item() <= 4e-2 * scale
            assert (x.float() - y.float()).abs().mean().item() <= 4e-3 * scale


def test_layer_norm_matches_torch():
    import torch.nn.functional as F
    import bevformer_tensorrt_amd.functions as hip_ops
    g = torch.Generator().manual_seed(0)
    for rows, C in [(40000, 256), (900, 256), (37, 64), (1001, 128), (513, 512), (1, 256)]:
        x = (torch.randn(rows, C, generator=g) * 3 + 0.5).half().cuda()
        w = torch.randn(C, generator=g).half().cuda()
        b = torch.randn(C, generator=g).half().cuda()
        want = F.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5)
        got = hip_ops.layer_norm(x, w, b, 1e-5)
        assert (got.float() - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
        got = hip_ops.layer_norm(x.view(1, rows, C), None, None, 1e-5)
        want = F.layer_norm(x.float(), (C,), None, None, 1e-5)
        assert got.shape == (1, rows, C)
        assert (got.float().view(rows, C) - want).abs().max().item() <= 2e-3 * max(1.0, want.abs().max().item())
        y = x.clone()
        hip_ops.layer_norm(y, w, b, 1e-5, out=y)     # in place
        assert torch.equal(y, hip_ops.layer_norm(x, w, b, 1e-5))


def test_int8_plugin_call_sites_track_fp16_model():
    """SURVEY.md 8f-2 / the model-level INT8 error budget: the tiny model with its MSDA / rotate call
    sites on the INT8 operators (entropy-calibrated scales from 3 frames) vs the fp16 operators on 2
    unseen frames.  Measured: bev_embed 2.0 % of its standard deviation, 98.5 % identical top-1 classes
    (profiles/r01f/int8_model_delta.jsonl); the bounds leave a 3x margin."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "int8_model_delta", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                         "int8_model_delta.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.run("tiny", 3, 2, "entropy")
    assert r["int8_sites"] == 49            # (3 x (TSA + SCA) + 6 decoder) MSDA sites x 4 tensors + rotate
    assert r["bev_embed_rel_err"] <= 0.06
    assert r["top1_class_agreement"] >= 0.93
    assert r["box_coord_mae"] <= 0.05


def _r3_ab(fn):
    from bevformer_tensorrt_amd import bevformer as B
    B._R3["enabled"] = True
    a = fn()
    B._R3["enabled"] = False
    try:
        b = fn()
    finally:
        B._R3["enabled"] = True
    return a, b


def test_r3_fusions_equal_module_by_module_path():
    """Round-3 launch-count work (split TSA projection, cached position terms, fused decoder attention,
    batched head, fused FPN / embedding passes) vs the same weights evaluated module by module."""
    import bevformer_tensorrt_amd.functions as hip_ops
    a, b = _r3_ab(lambda: run_sequence("tiny", hip_ops, torch.float16, n=3))
    for fa, fb in zip(a, b):
        for x, y in zip(fa, fb):
            scale = max(1.0, y.abs().max().item())
            assert (x - y).abs().max().item() <= 4e-2 * scale
            assert (x - y).abs().mean().item() <= 4e-3 * scale


def test_tsa_split_projection_equals_concatenated_linear():
    """TemporalSelfAttention with the stacked [192, 512] projection split along K (prev_bev | query + bev_pos,
    position term cached) vs cat + two Linear modules: fp16 rounding of the running sum only."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    dev, nq = torch.device("cuda"), 64 * 64
    torch.manual_seed(3)
    tsa = B.TemporalSelfAttention(hip_ops).to(dev, torch.float16)
    with torch.no_grad():
        tsa.sampling_offsets.weight.mul_(4.0)
    q = torch.randn(1, nq, 256, device=dev, dtype=torch.float16)
    prev = torch.randn(2, nq, 256, device=dev, dtype=torch.float16)
    pos = torch.randn(1, nq, 256, device=dev, dtype=torch.float16)
    ref = torch.rand(2, nq, 1, 2, device=dev, dtype=torch.float16)
    shapes = torch.tensor([[64, 64]])
    with torch.no_grad():
        (a1, a2), (b1, _) = _r3_ab(lambda: (tsa(q, prev, pos, ref, shapes), tsa(q, prev, pos, ref, shapes)))
    assert tsa._split is not None and tsa._pos_term is not None and tsa._pos_term[0][0] == pos.data_ptr()
    assert torch.equal(a1, a2)                      # the cached term is reused, not recomputed differently
    # ... also through a fresh VIEW of the same storage (the model passes bev_pos.view(1, nq, 256) every frame)
    term = tsa._pos_term[1]
    with torch.no_grad():
        a3 = tsa(q, prev, pos.view(1, nq, 256), ref, shapes)
    assert tsa._pos_term[1] is term and torch.equal(a1, a3)
    assert (a1.float() - b1.float()).abs().max().item() <= 2e-2 * max(1.0, b1.float().abs().max().item())
    assert (a1.float() - b1.float()).abs().mean().item() <= 2e-3


def test_decoder_layer_fused_attention_equals_modules():
    """DecoderLayer: one in-projection GEMM with the cached query_pos term + fused attention kernel + epilogue
    identities vs nn.MultiheadAttention and the separate adds."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    dev = torch.device("cuda")
    torch.manual_seed(4)
    layer = B.DecoderLayer(hip_ops).to(dev, torch.float16).eval()
    query = torch.randn(900, 1, 256, device=dev, dtype=torch.float16)
    qpos = torch.randn(900, 1, 256, device=dev, dtype=torch.float16)
    bev = torch.randn(2500, 1, 256, device=dev, dtype=torch.float16)
    ref = torch.rand(1, 900, 1, 2, device=dev, dtype=torch.float16)
    shapes = torch.tensor([[50, 50]])
    with torch.no_grad():
        a, b = _r3_ab(lambda: layer(query, bev, qpos, ref, shapes))
    assert layer._pos_qkv is not None and layer.cross_attn._pos_so is not None
    assert a.shape == b.shape == (900, 1, 256)
    assert (a.float() - b.float()).abs().max().item() <= 3e-2
    assert (a.float() - b.float()).abs().mean().item() <= 3e-3


def test_backbone_with_no_local_camera():
    """A rank of the camera-sharded path can own NO camera (6 cameras on 8 GPUs): backbone + neck on an empty image
    batch must run through every operator wrapper (empty in, empty out) and give the pyramid's shapes."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    dev = torch.device("cuda")
    for name in ("tiny", "small"):
        model = B.BEVFormer(name, ops=hip_ops, seed=0).to(dev, torch.float16)
        H, W = B.CONFIGS[name]["image"]
        img = torch.randn(1, 6, 3, H, W).to(dev, torch.float16)
        full = model.extract_feat(img, None)
        none = model.extract_feat(img, [])
        assert len(none) == len(full)
        for a, b in zip(none, full):
            assert a.shape[0] == 0 and tuple(a.shape[1:]) == tuple(b.shape[1:])
        one = model.extract_feat(img, [2])
        for a, b in zip(one, full):
            assert (a[0].float() - b[2].float()).abs().max().item() <= 3e-2 * max(1.0, b[2].float().abs().max().item())


class _NoCameraExchange:
    """Stand-in for camera_shard.CameraExchange on a rank that owns no camera, exchange "reduce" with nobody else."""
    mode, cams = "reduce", []

    def reduce(self, x):
        return x


def test_whole_frame_on_a_rank_without_cameras():
    """The full forward of the re-host for a rank that owns no camera (world 8, 6 cameras): empty backbone batch,
    empty value tensor in every SCA layer, zero partial sums into the exchange; finite outputs of the usual shapes."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer("small", ops=hip_ops, seed=0).to(dev, dtype)
    H, W = B.CONFIGS["small"]["image"]
    img = torch.randn(1, 6, 3, H, W).to(dev, dtype)
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    nq = model.bev_h * model.bev_w
    prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=dtype)
    can = torch.zeros(18, device=dev)
    bev, cls, crd = model(img, prev, torch.tensor(0.0, device=dev), can, l2i, [], _NoCameraExchange())
    assert bev.shape == (nq, 1, B.EMBED) and cls.shape[0] == 6 and crd.shape[-1] == 10
    assert torch.isfinite(bev.float()).all() and torch.isfinite(cls.float()).all() and torch.isfinite(crd.float()).all()


@pytest.mark.parametrize("cams", [None, [1, 4], [5]])
def test_base_frame_is_the_same_with_and_without_the_visibility_plan(cams):
    """BEVFormer-base, fp16: the frame with the SCA sampling on the per-rig visibility plan (round-5 default; for a
    camera-sharded rank the plan lists ITS cameras) against the same frame with the plan switched off (one block per
    1 280-query chunk, in-kernel compaction): identical BEV features and heads, bit for bit, over three frames with a
    calibration change in between (the plan is rebuilt with the projection).
    Under the rule-based dispatch (functions/linear.py: DETERMINISTIC -- every dense layer and convolution on the
    hand-written kernels), which is what a camera-sharded rank runs.  (Until round 6 the default dispatch was not
    run-to-run reproducible -- one library convolution and one library GEMM; see
    test_default_dispatch_frames_are_bit_reproducible.)"""
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    from bevformer_tensorrt_amd.functions import linear as Ln
    from bevformer_tensorrt_amd.functions import spatial_cross_attention as S

    class LocalOnly:            # the "reduce" exchange without a wire: this rank's masked camera sum is the result
        mode = "reduce"

        def __init__(self, cams):
            self.cams = cams

        def reduce(self, t):
            return t

    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer("base", seed=0).to(dev, dtype)
    H, W = B.CONFIGS["base"]["image"]
    l2i_a = G.synthetic_lidar2img((H, W)).to(dev)
    l2i_b = l2i_a.clone()
    l2i_b[:, :, 0, 3] += 3.0            # another rig: other visible sets
    outs = {}
    was = Ln.DETERMINISTIC["enabled"]
    Ln.DETERMINISTIC["enabled"] = True
    try:
        for planned in (True, False):
            S.PLANNED["enabled"] = planned
            r = B.FrameRunner(model, dev, dtype, cams=cams, gather=None if cams is None else LocalOnly(cams))
            got = []
            for i, (img, can, scene) in enumerate(frames((H, W), 3, dev, dtype)):
                cls, crd = r.step(img, can, l2i_a if i < 2 else l2i_b, scene)
                got.append((r.prev_bev.clone(), cls.clone(), crd.clone()))
            outs[planned] = got
    finally:
        S.PLANNED["enabled"] = True
        Ln.DETERMINISTIC["enabled"] = was
    for (ba, ca, da), (bb, cb, db) in zip(outs[True], outs[False]):
        assert torch.isfinite(ba.float()).all()
        assert torch.equal(ba, bb) and torch.equal(ca, cb) and torch.equal(da, db)


@pytest.mark.parametrize("name,repeats", [("small", 10), ("base", 5), ("tiny", 10)])
def test_default_dispatch_frames_are_bit_reproducible(name, repeats):
    """Round-5 review item 5: the same frame evaluated again under the DEFAULT dispatch gives the same bits, every
    hooked module and the three outputs.  Two launches used to break that (tools/probes/backbone_determinism.py,
    own_kernel_stress.py): MIOpen's split-reduction convolution behind the FPN's stride-2 level (the table now takes
    the hand-written implicit GEMM where it is within 5 %), and a hand-assembled stream-K GEMM of the library on the
    stage-1 conv3 of small (1 ulp on ~800 outputs in 0.4 % of its calls; the selection keeps to workspace-free,
    non-"Custom_" algorithms).  Behind the backbone the dense layers run on the hand-written kernels (_OWN_ENCODER).
    1 000 / 400 / 300 frames of small / tiny / base without a difference on the device box; here a handful."""
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer(name, seed=0).to(dev, dtype)
    H, W = B.CONFIGS[name]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, dtype)
    nq = model.bev_h * model.bev_w
    prev = (torch.randn(nq, 1, B.EMBED, generator=torch.Generator().manual_seed(2)) * 0.5).to(dev, dtype)
    can = torch.zeros(18, device=dev)
    can[0], can[-1] = 0.5, 0.8
    log = []
    for mod_name, mod in model.named_modules():
        if mod_name.count(".") == 1 and mod_name.split(".")[0] in ("encoder", "decoder"):
            mod.register_forward_hook(lambda m, i, o, n=mod_name: log.append((n, o.detach().clone())) if torch.is_tensor(o) else None)
    assert B._OWN_ENCODER["enabled"]
    first = None
    with torch.no_grad():
        model(img, prev, torch.tensor(1.0, device=dev), can, l2i)          # (first call: selects the library algorithms)
        for _ in range(repeats):
            log.clear()
            out = model(img, prev, torch.tensor(1.0, device=dev), can, l2i)
            got = list(log) + [("output%d" % i, t.clone()) for i, t in enumerate(out)]
            if first is None:
                first = got
                assert all(torch.isfinite(t.float()).all() for _, t in got)
                continue
            for (na, ta), (nb, tb) in zip(first, got):
                assert na == nb and torch.equal(ta, tb), "%s differs between two evaluations of the same frame" % na


def test_dense_layers_behind_the_backbone_run_on_the_hand_written_kernels():
    """_OWN_ENCODER (default): from the embeddings on -- the GEMMs that wrap the samplers, the decoder, the regression
    branches -- no dense layer of the base frame goes to hipBLASLt or the framework; the backbone keeps the shipped
    table's choice (library GEMMs where they measured faster)."""
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    from bevformer_tensorrt_amd.functions import linear as Ln
    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer("base", seed=0).to(dev, dtype)
    H, W = B.CONFIGS["base"]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    img = torch.randn(1, 6, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev, dtype)
    nq = model.bev_h * model.bev_w
    prev = torch.zeros(nq, 1, B.EMBED, device=dev, dtype=dtype)
    calls = {"backbone": {}, "transformer": {}}
    where = {"now": "backbone"}
    saved = dict(Ln._DENSE)
    for cand, fn in saved.items():
        def counted(*x, _c=cand, _f=fn, **k):
            calls[where["now"]][_c] = calls[where["now"]].get(_c, 0) + 1
            return _f(*x, **k)
        Ln._DENSE[cand] = counted
    lba = Ln.linear_bias_act       # (dense_auto's last resort when an own kernel declines a shape: must not happen here)

    def fallback(*x, **k):
        calls[where["now"]]["fallback"] = calls[where["now"]].get("fallback", 0) + 1
        return lba(*x, **k)
    Ln.linear_bias_act = fallback
    inner = model._transformer

    def transformer(*x, **k):
        where["now"] = "transformer"
        try:
            return inner(*x, **k)
        finally:
            where["now"] = "backbone"
    model._transformer = transformer
    try:
        with torch.no_grad():
            model(img, prev, torch.tensor(0.0, device=dev), torch.zeros(18, device=dev), l2i)
    finally:
        Ln._DENSE.update(saved)
        Ln.linear_bias_act = lba
    t = calls["transformer"]
    assert t.get("blaslt", 0) == 0 and t.get("torch", 0) == 0 and t.get("fallback", 0) == 0, t
    assert sum(t.values()) >= 6 * 6 + 6 * 8          # every encoder / decoder layer's dense layers went through the dispatch
    assert calls["backbone"].get("blaslt", 0) > 0     # ... while the backbone's 1x1 convolutions keep the table's choice


def test_fused_decoder_refinement_leaves_the_frame_bit_identical():
    """The one-launch reference-point refinement (bevops_refine_reference_points, default) against the framework's
    eight launches per decoder layer: the whole base frame -- BEV features, class logits, boxes -- bit for bit, over two
    frames (the second with history), eager and under graph replay."""
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer("base", seed=0).to(dev, dtype)
    H, W = B.CONFIGS["base"]["image"]
    l2i = G.synthetic_lidar2img((H, W)).to(dev)
    outs = {}
    try:
        for fused, graph in ((True, False), (False, False), (True, True)):
            B._FUSED_REFINE["enabled"] = fused
            r = B.FrameRunner(model, dev, dtype, graph=graph)
            got = []
            for img, can, scene in frames((H, W), 3, dev, dtype):
                cls, crd = r.step(img, can, l2i, scene)
                got.append((r.prev_bev.clone(), cls.clone(), crd.clone()))
            outs[(fused, graph)] = got
    finally:
        B._FUSED_REFINE["enabled"] = True
    for key in ((False, False), (True, True)):
        for (ba, ca, da), (bb, cb, db) in zip(outs[(True, False)], outs[key]):
            assert torch.equal(ba, bb) and torch.equal(ca, cb) and torch.equal(da, db), key


def _frame_metrics(got, want):
    """(bev_embed mean |err| / std of the reference, class-logit MAE, box MAE, top-1 class agreement of the last decoder
    layer) of one frame's (prev_bev, classes, boxes) against the reference evaluation's."""
    (bg, cg, dg), (bw, cw, dw) = got, want
    return (((bg - bw).abs().mean() / bw.std()).item(), (cg - cw).abs().mean().item(), (dg - dw).abs().mean().item(),
            (cg[-1].argmax(-1) == cw[-1].argmax(-1)).float().mean().item())


@pytest.mark.parametrize("name", ["small", "base"])
def test_fp16_product_frames_track_the_fp32_reference_formulation(name):
    """Model-level parity at the configs the bench quotes (BASELINE configs 3 and the headline): two frames (the first
    without history, the second with prev_bev, a can_bus shift and the rotate) of the fp16 PRODUCT path -- channels-last
    ResNet-101-DCN with the one-kernel stem, the DCNv2 implicit GEMM and the offset convolution, FPN, measured dense
    dispatch, fused TSA glue, for base the value projection into the sampler's planes + the planned hm5 sampler, frame
    replayed from its HIP graph with the camera projection and the plan build inside -- against THE SAME WEIGHTS
    evaluated module by module in fp32 with the reference's PyTorch formulations of the samplers (oracle/ref_ops.py:
    multi_scale_deformable_attn_pytorch, grid_sample rotate, DCNv2 in torch pinned to the C oracle; reference layout, no
    fusion, torch projection).  Dataflow: modules/transformer.py:245-398, modules/encoder.py:261-334.
    Measured (profiles/r06/model_parity.txt): small 0.32 / 0.35 % of the bev_embed standard deviation, class logits
    0.004, boxes 0.009, 97.7 / 96.7 % identical top-1 classes; base 0.57 / 0.65 %, 0.006, 0.011, 97.6 / 97.8 % (random
    weights: many near-ties between the ten class logits).  Bars, ~3 x the measured distance: bev_embed mean |err| <= 2 %
    of its standard deviation, class logits within 0.02 on average, boxes within 0.03, >= 95 % identical top-1 classes."""
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    from oracle.ref_ops import TorchRefOps
    dev = torch.device("cuda")
    H, W = B.CONFIGS[name]["image"]
    l2i = G.synthetic_lidar2img((H, W))

    def sequence(ops, dtype, graph):
        model = B.BEVFormer(name, ops=ops, seed=0).to(dev, dtype)
        runner = B.FrameRunner(model, dev, dtype, graph=graph)
        outs = []
        for k, (img, can, _) in enumerate(frames((H, W), 2, dev, torch.float32)):
            rig = l2i.clone()
            rig[:, :, :3, 3] += 0.25 * k                      # the calibration moves between the frames, as on nuScenes
            cls, crd = runner.step(img.to(dtype), can, rig, "scene")
            outs.append((runner.prev_bev.float().clone(), cls.float().clone(), crd.float().clone()))
        del model, runner
        torch.cuda.empty_cache()
        return outs

    want = sequence(TorchRefOps, torch.float32, False)
    got = sequence(hip_ops, torch.float16, True)
    for k, (g_, w_) in enumerate(zip(got, want)):
        assert all(torch.isfinite(t).all() for t in g_)
        rel, cls_mae, box_mae, top1 = _frame_metrics(g_, w_)
        print(f"{name} frame {k}: bev_embed rel err {rel:.4f}, class-logit MAE {cls_mae:.4f}, box MAE {box_mae:.4f}, "
              f"top-1 agreement {top1:.4f}")
        assert rel <= 0.02 and cls_mae <= 0.02 and box_mae <= 0.03 and top1 >= 0.95


def test_graph_replay_follows_the_calibration_of_every_frame():
    """lidar2img is a per-frame input (tools/bevformer/evaluate_trt.py:99,131-132): the frame's HIP graph evaluates the
    camera projection and the SCA visibility plan from the static calibration buffer on every replay.  Base config,
    reproducible dispatch: a graph runner fed the matrices [A, A', B] gives what an eager runner gives for the same
    sequence (other visible sets on frame 3), and NOT what the stale calibration would give."""
    from bevformer_tensorrt_amd import bevformer as B, geometry as G
    from bevformer_tensorrt_amd.functions import linear as Ln
    dev, dtype = torch.device("cuda"), torch.float16
    model = B.BEVFormer("base", seed=0).to(dev, dtype)
    H, W = B.CONFIGS["base"]["image"]
    a = G.synthetic_lidar2img((H, W))
    a2 = a.clone(); a2[:, :, :3, 3] += 0.05                 # frame-to-frame ego-motion jitter
    b = a.clone(); b[:, :, :2, :] *= 0.8                    # another rig (shorter focal lengths): other visible sets
    was = Ln.DETERMINISTIC["enabled"]
    Ln.DETERMINISTIC["enabled"] = True
    try:
        def run(rigs, graph):
            r = B.FrameRunner(model, dev, dtype, graph=graph)
            out = []
            for (img, can, _), rig in zip(frames((H, W), 3, dev, dtype), rigs):
                cls, crd = r.step(img, can, rig.clone(), "scene")      # a fresh tensor per frame, as the reference's loop
                out.append((r.prev_bev.float().clone(), cls.float().clone()))
            return out
        eager = run([a, a2, b], False)
        graph = run([a, a2, b], True)
        stale = run([a, a2, a2], True)
    finally:
        Ln.DETERMINISTIC["enabled"] = was
    for (be, ce), (bg, cg) in zip(eager, graph):
        scale = max(1.0, be.abs().max().item())
        assert (be - bg).abs().max().item() <= 4e-2 * scale and (be - bg).abs().mean().item() <= 4e-3 * scale
    d_true = (eager[2][0] - graph[2][0]).abs().mean().item()
    d_stale = (eager[2][0] - stale[2][0]).abs().mean().item()
    # the replay really used frame 3's matrices (measured with a 3-unit principal-point shift as the other rig: the
    # graph equals the eager frame bit for bit under the reproducible dispatch, d_true = 0, the frame on the stale
    # calibration 8e-4 away on average with 0.6 % of the BEV rows off by more than 1e-2)
    assert d_true <= 1e-5 and d_stale > 2e-4, (d_true, d_stale)
    changed = ((eager[2][0] - stale[2][0]).abs().amax(-1) > 1e-2).float().mean().item()
    assert changed > 0.01, changed             # ... and whole BEV rows differ, not rounding noise
