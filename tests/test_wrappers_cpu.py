"""L3 callers pinned to the reference.  The fixtures in
tests/golden/wrappers.npz are the outputs of the reference's OWN `forward_trt` methods
(SpatialCrossAttentionTRTP + MSDeformableAttention3DTRTP, TemporalSelfAttentionTRTP,
CustomMSDeformableAttentionTRTP, and the DetectionTransformerDecoderTRTP reference-point loop),
lifted from the reference tree by tests/golden/make_wrapper_golden.py and run with seeded
nn.Linear members.  The re-hosted modules of bevformer_tensorrt_amd/bevformer.py get the same
weights and inputs; here with the oracle operator (CPU), in tests/test_wrappers_gpu.py with the
HIP operators.  Bar: fp32, max abs <= 1e-5 (relative to the output scale ~1)."""
import numpy as np
import pytest
import torch

from conftest import golden
from util_refops import RefOps


def load(mod, g, prefix):
    with torch.no_grad():
        for name in ("value_proj", "sampling_offsets", "attention_weights", "output_proj"):
            lin = getattr(mod, name)
            lin.weight.copy_(torch.from_numpy(g[f"{prefix}.{name}.weight"]))
            lin.bias.copy_(torch.from_numpy(g[f"{prefix}.{name}.bias"]))
    return mod


def t(g, key, dev="cpu"):
    return torch.from_numpy(g[key]).to(dev)


def run_sca(ops, dev="cpu"):
    from bevformer_tensorrt_amd import bevformer as B
    g = golden("wrappers")
    levels = g["sca.shapes"]
    mod = load(B.SpatialCrossAttention(ops, levels=len(levels), points=8), g, "sca").to(dev)
    with torch.no_grad():
        out = mod(t(g, "sca.query", dev), t(g, "sca.value", dev), t(g, "sca.ref_cam", dev), t(g, "sca.bev_mask", dev),
                  torch.from_numpy(levels))
    return out.cpu().numpy(), g["sca.out"]


def run_tsa(ops, dev="cpu"):
    from bevformer_tensorrt_amd import bevformer as B
    g = golden("wrappers")
    mod = load(B.TemporalSelfAttention(ops, points=4), g, "tsa").to(dev)
    with torch.no_grad():
        out = mod(t(g, "tsa.query", dev), t(g, "tsa.prev", dev), t(g, "tsa.pos", dev), t(g, "tsa.ref_2d", dev),
                  torch.from_numpy(g["tsa.shapes"]))
    return out.cpu().numpy(), g["tsa.out"]


def run_dec(ops, dev="cpu"):
    from bevformer_tensorrt_amd import bevformer as B
    g = golden("wrappers")
    mod = load(B.CustomMSDeformableAttention(ops, points=4), g, "dec").to(dev)
    with torch.no_grad():
        out = mod(t(g, "dec.query", dev), t(g, "dec.bev", dev), t(g, "dec.query_pos", dev), t(g, "dec.ref", dev),
                  torch.from_numpy(g["dec.shapes"]))
    return out.cpu().numpy(), g["dec.out"]


@pytest.mark.parametrize("run", [run_sca, run_tsa, run_dec], ids=["sca", "tsa", "decoder_attn"])
def test_rehosted_wrapper_equals_reference_forward_trt(run):
    got, want = run(RefOps)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5, np.abs(got - want).max()


def test_decoder_reference_point_loop_bit_exact():
    """decoder.py:52-112: what each layer receives as reference_points and the refined points
    after each reg branch -- same torch ops in the same order => identical bits on the CPU."""
    from bevformer_tensorrt_amd import geometry as G
    g = golden("wrappers")
    ref = t(g, "loop.ref0")
    out = t(g, "loop.query")
    for i in range(3):
        layer_in = ref[..., :2].unsqueeze(2)
        assert np.array_equal(layer_in.numpy(), g["loop.layer_ref_in"][i])
        out = out + t(g, "loop.steps")[i]
        assert np.array_equal(out.numpy(), g["loop.inter"][i])
        tmp = torch.nn.functional.linear(out, t(g, f"loop.reg{i}.weight"), t(g, f"loop.reg{i}.bias")).view(1, -1, 10)
        ref = G.refine_reference_points(tmp, ref)
        assert np.array_equal(ref.numpy(), g["loop.inter_ref"][i]), i


def test_decoder_and_head_inverse_sigmoid_differ_only_below_eps():
    from bevformer_tensorrt_amd import geometry as G
    x = torch.tensor([0.0, 2e-6, 1e-5, 0.3, 1 - 1e-5, 1 - 2e-6, 1.0])
    a, b = G.inverse_sigmoid_decoder(x), G.inverse_sigmoid(x)
    assert torch.equal(a[2:5], b[2:5])
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    assert not torch.equal(a[:2], b[:2])
