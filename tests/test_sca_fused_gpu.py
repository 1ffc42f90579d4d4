"""GPU parity of the fused SCA sampling op: bevops_sca_forward must equal the
reference sequence  MSDA on the repeated query -> (queries * bev_mask).sum(0)  that it replaces
(det2trt/models/modules/spatial_cross_attention.py:254-270), evaluated with the fp32 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {
    # (num_cams, levels, nq, P, ppg)
    "tiny_sca": (6, [[15, 25]], 2500, 8, 4),
    "small_sca": (6, [[23, 40]], 22500, 8, 4),
    "base_sca_q4k": (6, [[116, 200], [58, 100], [29, 50], [15, 25]], 4000, 8, 4),
    "odd": (3, [[7, 9], [5, 3], [3, 1], [1, 1]], 2100, 4, 2),
    "ragged_chunk": (2, [[12, 17], [6, 9]], 2049 + 1280, 2, 1),
}


def gen(shape, seed=0, vis=0.3):
    ncam, levels, nq, P, ppg = shape
    heads, C = 8, 32
    g = torch.Generator().manual_seed(seed)
    L = len(levels)
    nk = sum(h * w for h, w in levels)
    value = torch.randn(ncam, nk, heads, C, generator=g)
    ref = torch.rand(ncam, nq, 1, 2 * ppg, generator=g) * 1.2 - 0.1
    off = torch.randn(1, nq, heads, L * P * 2, generator=g) * 1.5
    logit = torch.randn(1, nq, heads, L * P, generator=g)
    visible = torch.rand(ncam, nq, generator=g) < vis
    # visibility comes in runs along the BEV raster (like a camera frustum): whole runs on / off
    runs = torch.rand(ncam, (nq + 63) // 64, generator=g) < 0.5
    visible &= runs.repeat_interleave(64, dim=1)[:, :nq]
    count = visible.sum(0).clamp(min=1)
    mask = visible.float() / count                      # encoder.py:255-258 style weights
    sh = torch.tensor(levels, dtype=torch.int32)
    return [value.half().cuda(), sh.cuda(), ref.half().cuda(), off.half().cuda(), logit.half().cuda(),
            mask.half().cuda()]


@pytest.mark.parametrize("name", list(SHAPES))
def test_fused_sca_vs_oracle_composition(oracle_mod, name):
    import bevformer_tensorrt_amd as bev
    value, sh, ref, off, logit, mask = gen(SHAPES[name])
    ncam = value.shape[0]
    out = bev.spatial_cross_attention_sample(value, sh, ref, off, logit, mask.unsqueeze(-1))
    torch.cuda.synchronize()
    v, s, r, o, w = (a.float().cpu().numpy() if a.is_floating_point() else a.cpu().numpy()
                     for a in (value, sh, ref, off.expand(ncam, -1, -1, -1).contiguous(),
                               logit.expand(ncam, -1, -1, -1).contiguous()))
    q = oracle_mod.msda_f32(v, s, r, o, w).reshape(ncam, -1, 256)
    want = (q * mask.float().cpu().numpy()[:, :, None]).sum(0, keepdims=True)
    assert out.shape == (1, q.shape[1], 256)
    assert np.abs(out.float().cpu().numpy() - want).max() <= 1e-2


def test_fused_sca_equals_unfused_ops_and_ignores_masked_refs():
    """Against the op sequence on the GPU; and NaN reference points planted in masked-out
    (camera, query) pairs must not matter (those pairs are never read)."""
    import bevformer_tensorrt_amd as bev
    value, sh, ref, off, logit, mask = gen(SHAPES["base_sca_q4k"], seed=3)
    ncam = value.shape[0]
    queries = bev.multi_scale_deformable_attn(value, sh, ref, off.expand(ncam, -1, -1, -1),
                                              logit.expand(ncam, -1, -1, -1)).flatten(2)
    want = (queries.float() * mask.float().unsqueeze(-1)).sum(0, keepdim=True)
    got = bev.spatial_cross_attention_sample(value, sh, ref, off, logit, mask)
    assert (got.float() - want).abs().max().item() <= 4e-3
    ref2 = ref.clone()
    ref2[mask == 0] = float("nan")
    got2 = bev.spatial_cross_attention_sample(value, sh, ref2, off, logit, mask)
    assert torch.equal(got, got2)
    # all pairs masked out -> exact zeros
    z = bev.spatial_cross_attention_sample(value, sh, ref, off, logit, torch.zeros_like(mask))
    assert torch.count_nonzero(z).item() == 0


def test_new_entry_points_reject_bad_arguments():
    """Error behaviour of the additions to the C ABI: status codes, never a crash."""
    import ctypes
    from bevformer_tensorrt_amd.utils import lib as L
    h = L.load_library()
    value, sh, ref, off, logit, mask = gen(SHAPES["odd"])
    shapes_host = sh.cpu().contiguous()
    ncam, nk, heads, ch = value.shape
    nq = off.shape[1]
    out = torch.empty(1, nq, 256, dtype=torch.half, device="cuda")
    n = h.bevops_sca_workspace_size(L.F16, shapes_host.data_ptr(), ncam, nk, heads, ch, 4, nq, 4)
    assert n > 0
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    st = L.current_stream_ptr(value.device)
    args = lambda **kw: [kw.get("dtype", L.F16), value.data_ptr(), shapes_host.data_ptr(), ref.data_ptr(),
                         off.data_ptr(), logit.data_ptr(), kw.get("mask", mask.data_ptr()), out.data_ptr(), ncam,
                         kw.get("nk", nk), heads, ch, 4, nq, 4, 2, ws.data_ptr(), kw.get("bytes", n), st]
    assert h.bevops_sca_forward(*args()) == 0
    BAD, UNSUP = 2, 3                                              # include/bevops.h status codes
    assert h.bevops_sca_forward(*args(mask=None)) == BAD           # null pointer
    assert h.bevops_sca_forward(*args(nk=nk + 1)) == BAD           # level shapes do not add up to nk
    assert h.bevops_sca_forward(*args(bytes=n // 2)) == BAD        # workspace too small
    assert h.bevops_sca_forward(*args(dtype=L.F32)) == UNSUP
    assert h.bevops_sca_workspace_size(L.F32, shapes_host.data_ptr(), ncam, nk, heads, ch, 4, nq, 4) == 0
    x = torch.zeros(4, 24, dtype=torch.half, device="cuda")
    assert h.bevops_bias_act_nhwc(L.F16, x.data_ptr(), None, None, 4, 24, 1, st) == 0
    assert h.bevops_bias_act_nhwc(L.F16, x.data_ptr(), None, None, 4, 20, 1, st) == UNSUP  # channels % 8
    assert h.bevops_bias_act_nhwc(L.F16, x.data_ptr() + 2, None, None, 4, 24, 1, st) == BAD  # misaligned
    assert h.bevops_bias_act_nhwc(L.F32, x.data_ptr(), None, None, 4, 24, 1, st) == UNSUP
    torch.cuda.synchronize()


def test_projected_path_matches_projection_plus_fused_sampling():
    """bevops_value_proj_packed + bevops_sca_forward_prepacked (the value projection's GEMM epilogue writes the
    sampler's planes) against F.linear + spatial_cross_attention_sample on the same inputs: the planes hold the
    fp16 rounding of the same fp32 sums up to the GEMM's summation order, the sampling arithmetic is shared."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import geometry as G
    g = torch.Generator().manual_seed(0)
    levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
    nk = sum(h * w for h, w in levels)
    nq, heads, embed = 40000, 8, 256
    feats = (torch.randn(6, nk, embed, generator=g) * 0.5).half().cuda()
    wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
    bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
    ref3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
    cam, mask = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], G.synthetic_lidar2img((928, 1600)), (928, 1600))
    ref = cam.reshape(6, nq, 1, 8).half().cuda()
    bm = mask.reshape(6, nq, -1).any(-1).half().cuda()
    off = torch.randn(1, nq, heads, 64, generator=g).half().cuda()
    w = torch.randn(1, nq, heads, 32, generator=g).half().cuda()
    sh = torch.tensor(levels, dtype=torch.int32)
    value = torch.nn.functional.linear(feats, wgt, bias).view(6, nk, heads, 32)
    want = bev.spatial_cross_attention_sample(value, sh, ref, off, w, bm).float()
    got = bev.spatial_cross_attention_projected(feats, wgt, bias, sh, ref, off, w, bm, heads).float()
    assert torch.isfinite(got).all()
    err = (got - want).abs()
    assert err.max().item() <= 2e-2 and err.mean().item() <= 5e-4, (err.max().item(), err.mean().item())
    # determinism
    assert torch.equal(bev.spatial_cross_attention_projected(feats, wgt, bias, sh, ref, off, w, bm, heads).float(), got)


def test_packed_projection_planes_are_bit_identical_to_repacking_its_own_gemm():
    """The planes the value projection's GEMM epilogue writes (bevops_value_proj_packed) against the re-layout
    pass (bevops_value_pack_planes) applied to the SAME GEMM's row-major output (bevops_tsgemm_f16: identical
    accumulation order and rounding): every byte of the big and the staged set equal -- pads, pair partners
    across tile boundaries, level boundaries and cameras included."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    handle = L.load_library()
    g = torch.Generator().manual_seed(3)
    levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
    nk = sum(h * w for h, w in levels)
    ncam, heads, embed, nq, P = 6, 8, 256, 40000, 8
    feats = (torch.randn(ncam, nk, embed, generator=g) * 0.5).half().cuda()
    wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
    bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
    sh = torch.tensor(levels, dtype=torch.int32)
    st = L.current_stream_ptr(feats.device)
    nbytes = handle.bevops_value_proj_packed_size(sh.data_ptr(), ncam, nk, heads, 32, 4, nq, P)
    assert nbytes > 0
    a = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")     # poisoned: every byte must be written
    b = torch.full((nbytes,), 0xCD, dtype=torch.uint8, device="cuda")
    L.check(handle.bevops_value_proj_packed(feats.data_ptr(), wgt.data_ptr(), bias.data_ptr(), sh.data_ptr(), a.data_ptr(),
                                            nbytes, ncam, nk, heads, 32, 4, nq, P, st), "bevops_value_proj_packed")
    value = bev.tsgemm(feats.view(-1, embed), wgt, bias).view(ncam, nk, heads, 32).contiguous()
    L.check(handle.bevops_value_pack_planes(value.data_ptr(), sh.data_ptr(), b.data_ptr(), nbytes, ncam, nk, heads, 32, 4,
                                            nq, P, st), "bevops_value_pack_planes")
    torch.cuda.synchronize()
    # the planes = everything before the visibility bytes (the last bs * nq * heads bytes, rounded up to 256)
    planes = nbytes - ((ncam * nq * heads + 255) // 256) * 256
    diff = (a[:planes] != b[:planes])
    assert int(diff.sum()) == 0, (int(diff.sum()), int(diff.nonzero()[0]))


def _plan_case(kind, nq, seed):
    """Inputs of the projected SCA path at the base pyramid with a visibility pattern of the given kind."""
    from bevformer_tensorrt_amd import geometry as G
    g = torch.Generator().manual_seed(seed)
    levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
    nk = sum(h * w for h, w in levels)
    heads, embed = 8, 256
    feats = (torch.randn(6, nk, embed, generator=g) * 0.5).half().cuda()
    wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
    bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
    off = (torch.randn(1, nq, heads, 64, generator=g) * 2).half().cuda()
    w = torch.randn(1, nq, heads, 32, generator=g).half().cuda()
    if kind == "rig":
        assert nq == 40000
        ref3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
        cam, mask = G.point_sampling(ref3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], G.synthetic_lidar2img((928, 1600)), (928, 1600))
        ref = cam.reshape(6, nq, 1, 8)
        vis = mask.reshape(6, nq, -1).any(-1)
    else:
        ref = torch.rand(6, nq, 1, 8, generator=g) * 1.2 - 0.1
        if kind == "all":
            vis = torch.ones(6, nq, dtype=torch.bool)
        elif kind == "one_camera":          # five cameras see nothing: the slices of all blocks fall into camera 3
            vis = torch.zeros(6, nq, dtype=torch.bool)
            vis[3] = torch.rand(nq, generator=g) < 0.7
        elif kind == "few":                 # fewer visible pairs than blocks: most slices are empty
            vis = torch.zeros(6, nq, dtype=torch.bool)
            vis[0, 5] = vis[2, nq - 1] = vis[5, 0] = vis[5, 77] = True
        elif kind == "none":
            vis = torch.zeros(6, nq, dtype=torch.bool)
        else:                               # "random": uneven per-camera counts
            p = torch.tensor([0.05, 0.9, 0.3, 0.0, 0.5, 0.2]).view(6, 1)
            vis = torch.rand(6, nq, generator=g) < p
    bm = vis.float() / vis.sum(0).clamp(min=1)
    if kind == "weights":                   # weights that are not 1 / count: a lone camera with weight 0.5 is NOT "sole"
        bm = vis.float() * torch.tensor([1.0, 0.5, 0.25])[torch.randint(0, 3, vis.shape, generator=g)]
    bm = bm.half().cuda()
    sh = torch.tensor(levels, dtype=torch.int32)
    return feats, wgt, bias, sh, ref.half().cuda(), off, w, bm, heads


@pytest.mark.parametrize("kind,nq", [("rig", 40000), ("random", 40000), ("all", 9000), ("one_camera", 12345),
                                     ("few", 3000), ("none", 2500), ("random", 65535), ("weights", 20000)])
def test_planned_sampling_is_bit_identical_to_the_chunked_sampling(kind, nq):
    """bevops_sca_forward_planned (every block an equal slice of the visible (camera, query) pairs of a plan built
    from bev_mask by bevops_sca_plan_build) against bevops_sca_forward_prepacked (one block per 1 280-query chunk,
    in-kernel compaction): same arithmetic per pair, same camera reduction -> the same bits, whatever the visibility
    pattern and the number of slices per CU; with the pairs only one camera sees (weight exactly 1) stored by the
    sampler straight into the output rows (the default) and with every pair through the per-camera scratch."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    args = _plan_case(kind, nq, seed=nq % 97)
    bm = args[7]
    want = bev.spatial_cross_attention_projected(*args)
    plan = bev.spatial_cross_attention_plan(bm)
    assert plan is not None and plan.dtype == torch.uint8
    # the plan itself: counts, ascending lists (bits 0-15 of an entry) and the "only this camera, weight 1" bit (16)
    ncam = bm.shape[0]
    counts = plan[:64].view(torch.int32)[:ncam].cpu()
    pad = (nq + 63) // 64 * 64
    entries = plan[64:64 + ncam * pad * 4].view(torch.int32).view(ncam, pad).cpu()
    seen = (bm != 0).cpu()
    for c in range(ncam):
        vis_q = torch.nonzero(seen[c]).flatten()
        n = vis_q.numel()
        assert int(counts[c]) == n
        assert torch.equal(entries[c, :n] & 0xffff, vis_q.to(torch.int32))
        sole = (seen.sum(0) == 1) & (bm[c].cpu() == 1.0)
        assert torch.equal((entries[c, :n] >> 16) == 1, sole[vis_q]) and int((entries[c, :n] >> 17).abs().sum()) == 0
    handle = L.load_library()
    try:
        for direct in (3012, 3013):
            handle.bevops_msda_set_variant(direct)
            for k in (1, 2, 3):
                handle.bevops_msda_set_variant(3000 + k)
                got = bev.spatial_cross_attention_projected(*args, plan=plan)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (kind, direct, k, (got.float() - want.float()).abs().max().item())
        handle.bevops_msda_set_variant(3011)      # the rolled camera reduce (partner of the unrolled default)
        got = bev.spatial_cross_attention_projected(*args, plan=plan)
        assert torch.equal(got, want)
        for fold in (3015, 3014):                 # the round-5 build with broadcast moves / the folded default
            handle.bevops_msda_set_variant(fold)
            got = bev.spatial_cross_attention_projected(*args, plan=plan)
            assert torch.equal(got, want), (kind, fold)
        assert torch.equal(bev.spatial_cross_attention_projected(*args), want)      # ... and on the chunked path
    finally:
        handle.bevops_msda_set_variant(3002)      # the defaults: two slices per CU, direct stores, unrolled reduce
        handle.bevops_msda_set_variant(3012)
        handle.bevops_msda_set_variant(3010)
        handle.bevops_msda_set_variant(3014)
        handle.bevops_msda_set_variant(0)


def test_plan_entry_validates_its_arguments():
    from bevformer_tensorrt_amd.utils import lib as L
    h = L.load_library()
    assert h.bevops_sca_plan_size(6, 40000) == 64 + 6 * 40000 * 4 + 6 * 1024      # 40 000 is a multiple of 64; + scratch
    assert h.bevops_sca_plan_size(17, 100) == 0 and h.bevops_sca_plan_size(6, 65536) == 0
    m = torch.zeros(6, 100, dtype=torch.half, device="cuda")
    plan = torch.empty(h.bevops_sca_plan_size(6, 100), dtype=torch.uint8, device="cuda")
    st = L.current_stream_ptr(m.device)
    assert h.bevops_sca_plan_build(L.F16, m.data_ptr(), 6, 100, plan.data_ptr(), plan.numel() - 1, st) == L.BAD_PARAM
    assert h.bevops_sca_plan_build(L.F32, m.data_ptr(), 6, 100, plan.data_ptr(), plan.numel(), st) == L.NOT_SUPPORTED
    assert h.bevops_sca_plan_build(L.F16, m.data_ptr(), 6, 100, plan.data_ptr(), plan.numel(), st) == 0
    torch.cuda.synchronize()
    assert int(plan[:64].view(torch.int32)[:6].abs().sum()) == 0


def test_planned_sampling_on_the_rig_geometry_against_the_oracle_directly(oracle_mod):
    """The call the headline frame replays -- bevops_value_proj_packed + bevops_sca_forward_planned (hm5 kernel on the
    balanced slices of a visibility plan, single-camera pairs stored straight into the output, camera reduce for the
    others) -- on the REAL geometry: 40 000 BEV queries, reference points and bev_mask of the 6-camera rig from
    bevops_point_sampling, the plan from bevops_sca_plan_build.  Compared DIRECTLY with the oracle, no HIP <-> HIP hop:
    oracle.msda_f32 on the row-major fp16 value tensor the same GEMM produces (bevops_tsgemm_f16: the planes hold those
    very bits, test_packed_projection_planes_are_bit_identical_to_repacking_its_own_gemm), then the masked camera sum of
    spatial_cross_attention.py:254-270 in fp32.  Bars: north_star's fp16 tolerance, max |err| <= 1e-2, and mean |err| <=
    2e-4 (the bar of the drop-in op at full size, test_full_size_gpu.py)."""
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd import geometry as G
    g = torch.Generator().manual_seed(11)
    levels = [[116, 200], [58, 100], [29, 50], [15, 25]]
    nk = sum(h * w for h, w in levels)
    nq, heads, embed, ncam = 40000, 8, 256, 6
    feats = (torch.randn(ncam, nk, embed, generator=g) * 0.5).half().cuda()
    wgt = (torch.randn(embed, embed, generator=g) / 16).half().cuda()
    bias = (torch.randn(embed, generator=g) * 0.1).half().cuda()
    off = (torch.randn(1, nq, heads, 64, generator=g) * 2).half().cuda()       # ~2 px learned offsets at every level
    logit = torch.randn(1, nq, heads, 32, generator=g).half().cuda()
    ref_3d = G.reference_points_3d(200, 200, 8, 4, device="cpu")
    pillars = G.pillar_points(ref_3d, [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]).cuda()
    ref_cam, bev_mask = bev.point_sampling(pillars, G.synthetic_lidar2img((928, 1600)).cuda(), (928, 1600), torch.float16)
    ref = ref_cam.reshape(ncam, nq, 1, 8)
    sh = torch.tensor(levels, dtype=torch.int32)
    plan = bev.spatial_cross_attention_plan(bev_mask)
    got = bev.spatial_cross_attention_projected(feats, wgt, bias, sh, ref, off, logit, bev_mask, heads, plan=plan)
    torch.cuda.synchronize()
    assert got.shape == (1, nq, embed) and torch.isfinite(got.float()).all()
    # ---- the oracle, fed what the product was fed
    value = bev.tsgemm(feats.view(-1, embed), wgt, bias).view(ncam, nk, heads, 32)
    mask = bev_mask.float().cpu().numpy().reshape(ncam, nq)
    seen = mask != 0
    assert 0.1 < seen.mean() < 0.3 and (seen.sum(0) > 1).mean() > 0.01           # the rig: ~19 % visible, overlaps exist
    # reference points of invisible pairs may be +-inf in binary16 (pillars behind a camera): the reference's range gate
    # drops such samples and their pairs carry weight 0 anyway; the oracle gets finite stand-ins for them
    r = torch.nan_to_num(ref.float(), nan=-5.0, posinf=5.0, neginf=-5.0).cpu().numpy()
    q = oracle_mod.msda_f32(value.float().cpu().numpy(), sh.numpy(), r,
                            off.float().expand(ncam, -1, -1, -1).contiguous().cpu().numpy(),
                            logit.float().expand(ncam, -1, -1, -1).contiguous().cpu().numpy()).reshape(ncam, nq, embed)
    want = (q * mask[:, :, None]).sum(0, keepdims=True)
    err = np.abs(got.float().cpu().numpy() - want)
    print(f"planned SCA vs oracle, rig geometry: max {err.max():.3e} mean {err.mean():.3e}; |want| max {np.abs(want).max():.2f}")
    assert err.max() <= 1e-2
    assert err.mean() <= 2e-4
    # queries no camera sees: exact zeros
    assert not np.any(got.float().cpu().numpy()[0, ~seen.any(0)])
