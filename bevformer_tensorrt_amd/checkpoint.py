"""Loading the reference's checkpoints (the public BEVFormer / mmdet3d state dicts that
tools/bevformer/evaluate_pth.py:45 and the pth->onnx->trt flow start from) into the mmcv-free re-host.

The re-hosted modules (bevformer.py) carry the same parameters under shorter names and with every
frozen BatchNorm (`norm_eval=True`, configs/bevformer/bevformer_base.py:49) folded into the
convolution in front of it.  `reference_key_map(model)` lists, for every parameter of OUR model, the
reference key(s) it comes from; `load_reference_state_dict` applies it (folding the BN statistics)
and reports what was missing / left over.  Name sources in the reference tree:

  img_backbone.*      det2trt/models/backbones/resnet.py:106-260 (Bottleneck: conv1..3 + bn1..3 via
                      add_module(norm{1,2,3}_name), downsample = Sequential(conv, norm) from mmdet's
                      ResLayer), :507-509 (layer{i+1}); DCN pack: modules/cnn/dcn.py:31-86 (conv_offset)
  img_neck.*          third_party/bev_mmdet3d/models/necks/fpn.py:108-155 (lateral_convs / fpn_convs,
                      each an mmcv ConvModule -> `.conv`)
  pts_bbox_head.*     dense_heads/bevformer_head.py:121-209 (bev_embedding, query_embedding,
                      positional_encoding, cls_branches, reg_branches)
  ...transformer.*    modules/transformer.py:28-67 (level_embeds, cams_embeds, reference_points,
                      can_bus_mlp{0,2,norm}); encoder / decoder layers are mmcv BaseTransformerLayer
                      instances (attentions.{i}, ffns.0.layers.{0.0,1}, norms.{i}); SCA wraps its
                      sampler as `deformable_attention` (modules/spatial_cross_attention.py:58-104); the
                      decoder's self attention is mmcv MultiheadAttention (`attn` = nn.MultiheadAttention)
"""
import torch

BN_EPS = 1e-5   # mmcv build_norm_layer(dict(type="BN")) default


def _conv_bn(conv_key, bn_key):
    return ("conv_bn", conv_key, bn_key)


def reference_key_map(model):
    """{our parameter name: source}, source = a reference key (copied as is) or
    ("conv_bn", conv_prefix, bn_prefix) -> (weight, bias) of the convolution with the BN folded in.
    Folded entries are listed under OUR `<module>.weight`; the matching `<module>.bias` maps to the
    same tuple."""
    cfg = model.cfg
    m = {}

    def fold(ours, conv, bn):
        m[ours + ".weight"] = _conv_bn(conv, bn)
        m[ours + ".bias"] = _conv_bn(conv, bn)

    def copy(ours, ref, names=("weight", "bias")):
        for n in names:
            m[f"{ours}.{n}"] = f"{ref}.{n}"

    # ---- backbone
    fold("backbone.stem", "img_backbone.conv1", "img_backbone.bn1")
    for s, stage in enumerate(model.backbone.stages):
        for j, blk in enumerate(stage):
            ours, ref = f"backbone.stages.{s}.{j}", f"img_backbone.layer{s + 1}.{j}"
            fold(ours + ".conv1", ref + ".conv1", ref + ".bn1")
            fold(ours + ".conv2", ref + ".conv2", ref + ".bn2")
            if hasattr(blk.conv2, "conv_offset"):
                copy(ours + ".conv2.conv_offset", ref + ".conv2.conv_offset")
            fold(ours + ".conv3", ref + ".conv3", ref + ".bn3")
            if blk.downsample is not None:
                fold(ours + ".downsample", ref + ".downsample.0", ref + ".downsample.1")
    # ---- neck
    n_in = len(cfg["fpn_in"])
    for i in range(n_in):
        copy(f"neck.lateral.{i}", f"img_neck.lateral_convs.{i}.conv")
        copy(f"neck.fpn.{i}", f"img_neck.fpn_convs.{i}.conv")
    for k in range(len(model.neck.extra)):
        copy(f"neck.extra.{k}", f"img_neck.fpn_convs.{n_in + k}.conv")
    # ---- head
    H = "pts_bbox_head"
    m["bev_embedding.weight"] = f"{H}.bev_embedding.weight"
    m["query_embedding.weight"] = f"{H}.query_embedding.weight"
    m["row_embed.weight"] = f"{H}.positional_encoding.row_embed.weight"
    m["col_embed.weight"] = f"{H}.positional_encoding.col_embed.weight"
    for i in range(len(model.cls_branches)):
        for k in (0, 1, 3, 4, 6):     # Linear, LayerNorm, ReLU, Linear, LayerNorm, ReLU, Linear
            copy(f"cls_branches.{i}.{k}", f"{H}.cls_branches.{i}.{k}")
        for k in (0, 2, 4):           # Linear, ReLU, Linear, ReLU, Linear
            copy(f"reg_branches.{i}.{k}", f"{H}.reg_branches.{i}.{k}")
    # ---- transformer
    T = f"{H}.transformer"
    m["level_embeds"] = f"{T}.level_embeds"
    m["cams_embeds"] = f"{T}.cams_embeds"
    copy("reference_points", f"{T}.reference_points")
    copy("can_bus_mlp.0", f"{T}.can_bus_mlp.0")
    copy("can_bus_mlp.2", f"{T}.can_bus_mlp.2")
    copy("can_bus_mlp.4", f"{T}.can_bus_mlp.norm")
    att = ("sampling_offsets", "attention_weights", "value_proj", "output_proj")
    for i in range(len(model.encoder)):
        ours, ref = f"encoder.{i}", f"{T}.encoder.layers.{i}"
        for a in att:
            copy(f"{ours}.tsa.{a}", f"{ref}.attentions.0.{a}")
        for a in att[:3]:
            copy(f"{ours}.sca.{a}", f"{ref}.attentions.1.deformable_attention.{a}")
        copy(f"{ours}.sca.output_proj", f"{ref}.attentions.1.output_proj")
        copy(f"{ours}.ffn.fc1", f"{ref}.ffns.0.layers.0.0")
        copy(f"{ours}.ffn.fc2", f"{ref}.ffns.0.layers.1")
        for k in range(3):
            copy(f"{ours}.norms.{k}", f"{ref}.norms.{k}")
    for i in range(len(model.decoder)):
        ours, ref = f"decoder.{i}", f"{T}.decoder.layers.{i}"
        m[f"{ours}.self_attn.in_proj_weight"] = f"{ref}.attentions.0.attn.in_proj_weight"
        m[f"{ours}.self_attn.in_proj_bias"] = f"{ref}.attentions.0.attn.in_proj_bias"
        copy(f"{ours}.self_attn.out_proj", f"{ref}.attentions.0.attn.out_proj")
        for a in att:
            copy(f"{ours}.cross_attn.{a}", f"{ref}.attentions.1.{a}")
        copy(f"{ours}.ffn.fc1", f"{ref}.ffns.0.layers.0.0")
        copy(f"{ours}.ffn.fc2", f"{ref}.ffns.0.layers.1")
        for k in range(3):
            copy(f"{ours}.norms.{k}", f"{ref}.norms.{k}")
    return m


def fold_conv_bn(weight, conv_bias, gamma, beta, mean, var, eps=BN_EPS):
    """Frozen BatchNorm behind a convolution as the convolution's own scale and shift (float64)."""
    w = weight.double()
    scale = gamma.double() / torch.sqrt(var.double() + eps)
    shift = beta.double() - mean.double() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.double() * scale
    return w * scale.view(-1, *([1] * (w.dim() - 1))), shift


def load_reference_state_dict(model, state_dict, strict=True):
    """Fill `model` (bevformer.BEVFormer) from a reference checkpoint's state dict (the value of its
    "state_dict" entry, or the dict itself).  Returns (missing, unexpected): our parameters that found
    no source, and reference tensors nothing asked for (BN `num_batches_tracked` and the head's
    `code_weights` buffer are expected leftovers and not reported).  strict: raise on either."""
    sd = state_dict.get("state_dict", state_dict)
    kmap = reference_key_map(model)
    own = dict(model.named_parameters())
    used, missing = set(), []
    with torch.no_grad():
        for name, p in own.items():
            src = kmap.get(name)
            if src is None:
                missing.append(name)
                continue
            if isinstance(src, tuple):
                _, conv, bn = src
                keys = [conv + ".weight", bn + ".weight", bn + ".bias", bn + ".running_mean", bn + ".running_var"]
                if any(k not in sd for k in keys):
                    missing.append(name)
                    continue
                w, b = fold_conv_bn(sd[keys[0]], sd.get(conv + ".bias"), *(sd[k] for k in keys[1:]))
                used.update(keys)
                if conv + ".bias" in sd:
                    used.add(conv + ".bias")
                val = w if name.endswith(".weight") else b
            else:
                if src not in sd:
                    missing.append(name)
                    continue
                used.add(src)
                val = sd[src]
            if name == "level_embeds" and val.dim() == 2 and val.shape[0] > p.shape[0] and val.shape[1:] == p.shape[1:]:
                # tiny / small configs never pass num_feature_levels, so PerceptionTransformer allocates its default
                # 4 rows (modules/transformer.py:15,55) although only level_embeds[lvl] for lvl < the FPN's levels is
                # ever read (:313-321): take the rows that are used
                val = val[:p.shape[0]]
            if tuple(val.shape) != tuple(p.shape):
                raise ValueError(f"{name}: shape {tuple(p.shape)} vs reference {tuple(val.shape)}")
            p.copy_(val.to(p.dtype))
    ignorable = ("num_batches_tracked", "code_weights")
    unexpected = sorted(k for k in sd if k not in used and not k.endswith(ignorable))
    if strict and (missing or unexpected):
        raise KeyError(f"missing {missing[:8]}{'...' if len(missing) > 8 else ''}; "
                       f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    model._static = None          # cached geometry / packed weights are rebuilt on the next frame
    return missing, unexpected
