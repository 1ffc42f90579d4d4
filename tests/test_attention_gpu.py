"""GPU parity of the decoder's self-attention kernel (csrc/attention.hip) against the fp32 evaluation of the same fp16
operands (torch scaled_dot_product_attention in fp32): 900 object queries x 8 heads x 32 (decoder.py:52-112), ragged
query counts (partial query and key blocks), one head, large logits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,heads,gain", [(900, 8, 1.0), (900, 8, 6.0), (1, 8, 1.0), (31, 2, 1.0), (33, 1, 2.0),
                                          (128, 8, 1.0), (129, 4, 1.0), (1024, 8, 1.0), (517, 3, 3.0)])
def test_self_attention_matches_fp32_reference(n, heads, gain):
    import bevformer_tensorrt_amd as bev
    g = torch.Generator().manual_seed(n + heads)
    qkv = torch.randn(n, 3, heads, 32, generator=g)
    qkv[:, :2] *= gain                   # sharper softmax: tests the running maximum across key blocks
    qkv = qkv.half().cuda()
    got = bev.self_attention_qkv(qkv)
    assert got.shape == (n, heads * 32) and got.dtype == torch.float16
    q, k, v = (qkv[:, i].float().transpose(0, 1) for i in range(3))            # [heads, n, 32]
    want = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].transpose(0, 1).reshape(n, heads * 32)
    err = (got.float() - want).abs()
    assert torch.isfinite(got.float()).all()
    assert err.max().item() <= 4e-3 * max(1.0, want.abs().max().item()) and err.mean().item() <= 3e-4, \
        (err.max().item(), err.mean().item())
    assert torch.equal(bev.self_attention_qkv(qkv), got)         # deterministic


def test_self_attention_domain():
    import bevformer_tensorrt_amd as bev
    from bevformer_tensorrt_amd.utils import lib as L
    with pytest.raises(L.BevopsError) as e:
        bev.self_attention_qkv(torch.zeros(64, 3, 8, 64, dtype=torch.half, device="cuda"))
    assert e.value.status == L.NOT_SUPPORTED
    with pytest.raises(L.BevopsError) as e:
        bev.self_attention_qkv(torch.zeros(1025, 3, 8, 32, dtype=torch.half, device="cuda"))
    assert e.value.status == L.NOT_SUPPORTED


def test_decoder_layer_with_own_attention_equals_framework_attention():
    import bevformer_tensorrt_amd.functions as hip_ops
    from bevformer_tensorrt_amd import bevformer as B
    dev = torch.device("cuda")
    torch.manual_seed(4)
    layer = B.DecoderLayer(hip_ops).to(dev, torch.float16).eval()
    query = torch.randn(900, 1, 256, device=dev, dtype=torch.float16)
    qpos = torch.randn(900, 1, 256, device=dev, dtype=torch.float16)
    bev_embed = torch.randn(2500, 1, 256, device=dev, dtype=torch.float16)
    ref = torch.rand(1, 900, 1, 2, device=dev, dtype=torch.float16)
    shapes = torch.tensor([[50, 50]])
    with torch.no_grad():
        a = layer(query, bev_embed, qpos, ref, shapes)
        B._OWN_ATTN["enabled"] = False
        try:
            b = layer(query, bev_embed, qpos, ref, shapes)
        finally:
            B._OWN_ATTN["enabled"] = True
    assert (a.float() - b.float()).abs().max().item() <= 2e-2 and (a.float() - b.float()).abs().mean().item() <= 1e-3
