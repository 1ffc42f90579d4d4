"""Checkpoint key map (bevformer_tensorrt_amd/checkpoint.py): every parameter of the re-hosted model has
exactly one source in the reference's state-dict naming, the names are the reference's own module
attributes (fixture extracted from the reference tree by AST, tests/golden/make_keymap_golden.py), frozen
BatchNorms fold exactly, and a synthetic reference-style state dict round-trips."""
import json
import os

import pytest
import torch

from bevformer_tensorrt_amd import bevformer as B
from bevformer_tensorrt_amd import checkpoint as C

NAMES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_module_names.json")))


def cls_names(suffix):
    hits = [v["names"] for k, v in NAMES.items() if k.endswith("::" + suffix)]
    assert hits, suffix
    return set(hits[0])


@pytest.fixture(scope="module", params=["tiny", "small", "base"])
def model(request):
    return B.BEVFormer(request.param, ops=object())   # no operator is called here


def test_every_parameter_has_exactly_one_source(model):
    kmap = C.reference_key_map(model)
    own = [n for n, _ in model.named_parameters()]
    assert sorted(own) == sorted(kmap), (set(own) ^ set(kmap))
    plain = [v for v in kmap.values() if isinstance(v, str)]
    assert len(plain) == len(set(plain))                       # no reference tensor feeds two parameters
    folded = {}
    for k, v in kmap.items():
        if isinstance(v, tuple):
            folded.setdefault(v, []).append(k)
    # a folded convolution fills exactly its weight and its bias
    assert all(sorted(x.rsplit(".", 1)[1] for x in ks) == ["bias", "weight"] for ks in folded.values())


def test_names_are_the_reference_modules_attributes(model):
    """Path components that the reference tree itself defines (mmcv's own containers -- attentions, ffns,
    norms, layers, ConvModule.conv, MultiheadAttention.attn, bn{1,2,3} -- are outside the tree)."""
    head, trans = cls_names("BEVFormerHeadTRT"), cls_names("PerceptionTransformerTRT")
    tsa, sca = cls_names("TemporalSelfAttentionTRT"), cls_names("SpatialCrossAttentionTRT")
    msda3d, dec = cls_names("MSDeformableAttention3DTRT"), cls_names("CustomMSDeformableAttentionTRT")
    bott, fpn = cls_names("Bottleneck"), cls_names("CustomFPN")
    dcn = cls_names("ModulatedDeformConv2dPackPlugin")
    for ours, src in C.reference_key_map(model).items():
        keys = [src] if isinstance(src, str) else [src[1], src[2]]
        for key in keys:
            p = key.split(".")
            if p[0] == "img_backbone" and p[1].startswith("layer"):
                assert p[3] in bott | {"bn1", "bn2", "bn3"}, key
                if "conv_offset" in p:
                    assert "conv_offset" in dcn
            elif p[0] == "img_neck":
                assert p[1] in fpn, key
            elif p[0] == "pts_bbox_head" and p[1] == "transformer":
                if p[2] in ("encoder", "decoder"):
                    assert p[2] in trans
                    leaf = p[-2]
                    if p[5] == "attentions":
                        if p[2] == "encoder" and p[6] == "0":
                            assert leaf in tsa, key
                        elif p[2] == "encoder" and "deformable_attention" in p:
                            assert "deformable_attention" in sca and leaf in msda3d, key
                        elif p[2] == "encoder":
                            assert leaf in sca, key
                        elif p[6] == "1":
                            assert leaf in dec, key
                else:
                    assert p[2] in trans or ".".join(p[2:4]) in trans, key
            elif p[0] == "pts_bbox_head" and p[1] != "positional_encoding":   # positional_encoding: mmdet DETRHead
                assert p[1] in head, key


def test_fold_conv_bn_is_exact():
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(5, 7, 3, 1, 1, bias=False).double()
    bn = torch.nn.BatchNorm2d(7).double().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.2, 2.0)
    x = torch.randn(2, 5, 9, 8, dtype=torch.float64)
    w, b = C.fold_conv_bn(conv.weight, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    want = bn(conv(x))
    got = torch.nn.functional.conv2d(x, w, b, 1, 1)
    assert (got - want).abs().max().item() < 1e-12


def synthetic_reference_state_dict(model, seed=0):
    """A state dict in the reference's naming whose folded form equals `model`'s current parameters:
    random BN statistics, convolution weights un-folded accordingly."""
    g = torch.Generator().manual_seed(seed)
    own = dict(model.named_parameters())
    sd, done = {}, set()
    for name, src in C.reference_key_map(model).items():
        if isinstance(src, str):
            sd[src] = own[name].detach().clone()
        elif src not in done:
            done.add(src)
            _, conv, bn = src
            mod = name.rsplit(".", 1)[0]
            w, b = own[mod + ".weight"].detach().double(), own[mod + ".bias"].detach().double()
            c = w.shape[0]
            gamma = torch.rand(c, generator=g).double() + 0.5
            var = torch.rand(c, generator=g).double() + 0.3
            mean = torch.randn(c, generator=g).double()
            scale = gamma / torch.sqrt(var + C.BN_EPS)
            sd[conv + ".weight"] = (w / scale.view(-1, 1, 1, 1)).float()
            sd[bn + ".weight"], sd[bn + ".running_var"], sd[bn + ".running_mean"] = gamma.float(), var.float(), mean.float()
            sd[bn + ".bias"] = (b + mean * scale).float()
            sd[bn + ".num_batches_tracked"] = torch.tensor(1)
    sd["pts_bbox_head.code_weights"] = torch.ones(10)
    return sd


def test_round_trip_tiny():
    a = B.BEVFormer("tiny", ops=object(), seed=1)
    with torch.no_grad():       # give the (zero-initialised) shifts some content
        for n, p in a.named_parameters():
            if n.endswith("bias"):
                p.normal_(std=0.1)
    sd = synthetic_reference_state_dict(a)
    b = B.BEVFormer("tiny", ops=object(), seed=2)
    missing, unexpected = C.load_reference_state_dict(b, {"state_dict": sd})
    assert missing == [] and unexpected == []
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    worst = max((pa[k] - pb[k]).abs().max().item() / (pa[k].abs().max().item() + 1e-6) for k in pa)
    assert worst < 1e-5
    # strictness: a missing tensor and a stray one are both reported
    sd2 = dict(sd)
    sd2.pop("pts_bbox_head.transformer.level_embeds")
    sd2["pts_bbox_head.some_new_buffer"] = torch.zeros(1)
    with pytest.raises(KeyError):
        C.load_reference_state_dict(b, sd2)
    missing, unexpected = C.load_reference_state_dict(b, sd2, strict=False)
    assert missing == ["level_embeds"] and unexpected == ["pts_bbox_head.some_new_buffer"]


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_reference_shaped_level_embeds_load(name):
    """The reference's tiny / small configs never pass num_feature_levels, so their checkpoints carry a [4, 256]
    level_embeds (modules/transformer.py:15,55) although one FPN level is used: the rows that are read are taken."""
    a = B.BEVFormer(name, ops=object(), seed=1)
    sd = synthetic_reference_state_dict(a)
    key = "pts_bbox_head.transformer.level_embeds"
    assert sd[key].shape == (1, 256)
    ref = torch.randn(4, 256)
    ref[0] = sd[key][0]
    sd[key] = ref                                   # the reference's shape
    b = B.BEVFormer(name, ops=object(), seed=2)
    missing, unexpected = C.load_reference_state_dict(b, sd)
    assert missing == [] and unexpected == []
    assert torch.equal(b.level_embeds.detach(), ref[:1])
    sd[key] = torch.randn(4, 128)                   # a genuinely different shape still fails loudly
    with pytest.raises(ValueError):
        C.load_reference_state_dict(b, sd)


def test_tiny_backbone_is_pytorch_style_and_base_caffe():
    """configs/bevformer/bevformer_tiny.py:62 (style="pytorch": stride on the 3x3), bevformer_base.py:50."""
    t = B.BEVFormer("tiny", ops=object()).backbone.stages[1][0]
    assert t.conv1.stride == (1, 1) and t.conv2.stride == (2, 2)
    s = B.BEVFormer("small", ops=object()).backbone.stages[1][0]
    assert s.conv1.stride == (2, 2) and s.conv2.stride == (1, 1)
