"""oracle/detr_oracle.py -- CPU ORACLE for the detector (test infrastructure, NOT product code).

Plain torch-functional fp32 restatement of the reference detector's inference path, one function per
reference module, taking the reference's own state_dict.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this file.

Reference citations (paths relative to /root/reference):
  frozen_bn ............ src/models/backbone.py:46-56
  resnet_body .......... torchvision.models.resnet50/101 (ResNet v1.5 bottleneck: stride on the 3x3,
                         blocks [3,4,6,3] / [3,4,23,3]) as instantiated at src/models/backbone.py:90-92;
                         torchvision itself is not vendored in the reference and not installed here, so the
                         ResNet body cannot be pinned against a reference run: the public
                         architecture is restated and pinned against an independent third-party implementation
                         of it instead (transformers.ResNetModel with the same weights, tests/test_resnet_pin.py:
                         layer4 features agree to 2e-5 of their range for R50 and R101).  Everything else below
                         is pinned against the imported reference (tests/golden/make_golden_detr.py).
  position_embedding ... src/models/position_encoding.py:26-46 (normalize=True, 128 feats/axis)
  transformer .......... src/models/transformer.py:45-57, 154-167, 217-238, 107-129
  detr_forward ......... src/models/detr.py:49-94
  postprocess / nms_3d . src/models/detr.py:96-205, src/utils/box_utils.py:123-144, 424-447, 490-494
  detection rows ....... src/processor.py:259-289
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def frozen_bn(x, sd, p):
    w = sd[p + ".weight"].reshape(1, -1, 1, 1)
    b = sd[p + ".bias"].reshape(1, -1, 1, 1)
    rv = sd[p + ".running_var"].reshape(1, -1, 1, 1)
    rm = sd[p + ".running_mean"].reshape(1, -1, 1, 1)
    scale = w * (rv + 1e-5).rsqrt()
    bias = b - rm * scale
    return x * scale + bias


def resnet_body(x, sd, blocks=(3, 4, 6, 3), prefix="backbone.0.body.", dilation=False):
    """torchvision ResNet-50/101 v1.5 body (absent from the reference tree; restated from its published definition).  `dilation`:
    replace_stride_with_dilation = [False, False, True] (src/models/backbone.py:89-91) -- layer4 with stride 1, its 3x3 filters
    dilated by 2 (padding 2) from the second block on, the first block at the previous dilation (ResNet._make_layer)."""
    x = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, prefix + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for l in range(4):
        for i in range(blocks[l]):
            p = f"{prefix}layer{l + 1}.{i}."
            dc5 = dilation and l == 3
            stride = 2 if (i == 0 and l > 0 and not dc5) else 1
            dil = 2 if (dc5 and i > 0) else 1
            idt = x
            out = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
            out = F.relu(frozen_bn(F.conv2d(out, sd[p + "conv2.weight"], None, stride=stride, padding=dil, dilation=dil), sd, p + "bn2"))
            out = frozen_bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
            if i == 0:
                idt = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride), sd, p + "downsample.1")
            x = F.relu(out + idt)
    return x


def position_embedding(h, w, num_pos_feats=128, temperature=10000, batch=1):
    not_mask = torch.ones(batch, h, w, dtype=torch.bool)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)  # [B, 256, h, w]


def _mha(sd, p, q, k, v, nheads):
    # nn.MultiheadAttention.forward -> F.multi_head_attention_forward, need_weights=True (explicit softmax path)
    E = q.shape[-1]
    out, _ = F.multi_head_attention_forward(
        q, k, v, E, nheads, sd[p + "in_proj_weight"], sd[p + "in_proj_bias"], None, None, False, 0.0,
        sd[p + "out_proj.weight"], sd[p + "out_proj.bias"], training=False, key_padding_mask=None, need_weights=True)
    return out


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def position_embedding_learned(sd, h, w, batch=1, prefix="backbone.1."):
    """PositionEmbeddingLearned.forward (src/models/position_encoding.py:49-74) -> [B, C, h, w]"""
    x_emb = sd[prefix + "col_embed.weight"][torch.arange(w)]
    y_emb = sd[prefix + "row_embed.weight"][torch.arange(h)]
    pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)
    return pos.permute(2, 0, 1).unsqueeze(0).repeat(batch, 1, 1, 1)


def _ffn(x, sd, p):
    return F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])


def transformer_pre(src, pos, query_embed, sd, nheads=8, enc_layers=6, dec_layers=6, prefix="transformer."):
    """`normalize_before` variant (src/models/transformer.py: encoder layer forward_pre :169-183, final encoder norm :26-28 /
    :82-83, decoder layer forward_pre :240-262, per-layer decoder norm :117-127): same shapes as `transformer`."""
    bs = src.shape[0]
    x = src.flatten(2).permute(2, 0, 1)
    pe = pos.flatten(2).permute(2, 0, 1)
    qe = query_embed.unsqueeze(1).repeat(1, bs, 1)
    for i in range(enc_layers):
        p = f"{prefix}encoder.layers.{i}."
        x2 = _ln(x, sd, p + "norm1")
        q = k = x2 + pe
        x = x + _mha(sd, p + "self_attn.", q, k, x2, nheads)
        x = x + _ffn(_ln(x, sd, p + "norm2"), sd, p)
    memory = _ln(x, sd, prefix + "encoder.norm")
    tgt = torch.zeros_like(qe)
    inter = []
    for i in range(dec_layers):
        p = f"{prefix}decoder.layers.{i}."
        t2 = _ln(tgt, sd, p + "norm1")
        q = k = t2 + qe
        tgt = tgt + _mha(sd, p + "self_attn.", q, k, t2, nheads)
        t2 = _ln(tgt, sd, p + "norm2")
        tgt = tgt + _mha(sd, p + "multihead_attn.", t2 + qe, memory + pe, memory, nheads)
        tgt = tgt + _ffn(_ln(tgt, sd, p + "norm3"), sd, p)
        inter.append(_ln(tgt, sd, prefix + "decoder.norm"))
    return torch.stack(inter).transpose(1, 2), memory


def transformer(src, pos, query_embed, sd, nheads=8, enc_layers=6, dec_layers=6, prefix="transformer.", pre_norm=False):
    """src [B,C,h,w], pos [B,C,h,w], query_embed [Q,C] -> hs [dec_layers,B,Q,C], memory [HW,B,C]"""
    if pre_norm:
        return transformer_pre(src, pos, query_embed, sd, nheads, enc_layers, dec_layers, prefix)
    bs = src.shape[0]
    x = src.flatten(2).permute(2, 0, 1)
    pe = pos.flatten(2).permute(2, 0, 1)
    qe = query_embed.unsqueeze(1).repeat(1, bs, 1)
    for i in range(enc_layers):
        p = f"{prefix}encoder.layers.{i}."
        q = k = x + pe
        x = _ln(x + _mha(sd, p + "self_attn.", q, k, x, nheads), sd, p + "norm1")
        ff = F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                      sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = _ln(x + ff, sd, p + "norm2")
    memory = x
    tgt = torch.zeros_like(qe)
    inter = []
    for i in range(dec_layers):
        p = f"{prefix}decoder.layers.{i}."
        q = k = tgt + qe
        tgt = _ln(tgt + _mha(sd, p + "self_attn.", q, k, tgt, nheads), sd, p + "norm1")
        tgt = _ln(tgt + _mha(sd, p + "multihead_attn.", tgt + qe, memory + pe, memory, nheads), sd, p + "norm2")
        ff = F.linear(F.relu(F.linear(tgt, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                      sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        tgt = _ln(tgt + ff, sd, p + "norm3")
        inter.append(_ln(tgt, sd, prefix + "decoder.norm"))
    return torch.stack(inter).transpose(1, 2), memory


def _mlp(x, sd, p, n=3):
    for i in range(n):
        x = F.linear(x, sd[f"{p}.layers.{i}.weight"], sd[f"{p}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return x


def heads(hs, sd):
    return {
        "pred_logits": F.linear(hs, sd["class_embed.weight"], sd["class_embed.bias"]),
        "pred_boxes": _mlp(hs, sd, "bbox_embed").sigmoid(),
        "pred_angle": _mlp(hs, sd, "angle_embed"),
        "pred_offset": _mlp(hs, sd, "offset_embed"),
        "pred_size": _mlp(hs, sd, "size_embed"),
        "pred_depth": _mlp(hs, sd, "depth_embed"),
        "pred_obj_features": hs,
    }


@torch.no_grad()
def detr_forward(sd, img, blocks=(3, 4, 6, 3), nheads=8, enc_layers=6, dec_layers=6, return_taps=False, pre_norm=False, learned_pos=False,
                 dilation=False):
    """img [B,3,H,W] float32 CPU -> the reference's output dict for the last decoder layer."""
    feat = resnet_body(img, sd, blocks, dilation=dilation)
    B, _, h, w = feat.shape
    pos = position_embedding_learned(sd, h, w, batch=B) if learned_pos else position_embedding(h, w, batch=B)
    src = F.conv2d(feat, sd["input_proj.weight"], sd["input_proj.bias"])
    hs, memory = transformer(src, pos, sd["query_embed.weight"], sd, nheads, enc_layers, dec_layers, pre_norm=pre_norm)
    out = heads(hs[-1], sd)
    if return_taps:
        out["_layer4"] = feat
        out["_memory"] = memory.permute(1, 0, 2)  # [B, HW, C]
    return out


# ---- bf16-faithful mode (BASELINE config 4) ---------------------------------------------------------------------------
# The same network with every value rounded to bf16 (nearest even) exactly where the gfx950 bf16 mode stores one
# (odam_amd/csrc/detr_model.hip forward_impl with dtype 1): the image, every weight matrix, every conv / linear output
# after its fp32 epilogue (x scale + bias (+ residual) (ReLU)), the attention probabilities as the PV operand and the
# attention output, both LayerNorm outputs (y and y + pos, each rounded from the fp32 value).  Accumulation, FrozenBN
# scale / bias, biases, softmax (per 64-key tile, running max, exp2 with the scale folded in: attention_bf16_kernel) and
# LayerNorm statistics are fp32, the head outputs are not rounded.  What still differs from the kernel is the order of
# fp32 summation inside a contraction, i.e. a last-bit difference that occasionally flips one bf16 rounding.
# The reference itself has no bf16 mode (src/models/detr.py:49-94 runs fp32); this restatement is the checker of OUR bf16
# mode, so that config 4 can be held to a tight tolerance instead of "close to fp32".
def _rb(x):
    return x.to(torch.bfloat16).to(torch.float32)


# Teacher forcing (tests/test_detr_gpu.py::test_bf16_layers_teacher_forced): when TRACE is a list, every contraction of
# detr_forward_bf16 appends its operands and its result, so that the kernel can be run layer by layer on the oracle's own
# inputs.  Rounding is chaotic -- once two evaluations differ by a fraction of a bf16 ulp their later roundings are
# independent -- so over the ~120 layers of a forward the kernel and this restatement decorrelate to the bf16 noise level
# itself; layer by layer, on identical inputs, they must agree bit for bit up to rare one-ulp ties of the fp32 summation order.
TRACE = None


def _bn_fold(sd, p):
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
    return scale, sd[p + ".bias"] - sd[p + ".running_mean"] * scale


def _conv_b(x, sd, wkey, bn=None, bias=None, stride=1, padding=0, res=None, relu=False, out_f32=False):
    y = F.conv2d(x, _rb(sd[wkey]), None, stride=stride, padding=padding)
    if bn is not None:
        sc, bi = _bn_fold(sd, bn)
        y = y * sc.reshape(1, -1, 1, 1) + bi.reshape(1, -1, 1, 1)
    elif bias is not None:
        y = y + bias.reshape(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    y = y if out_f32 else _rb(y)
    if TRACE is not None:
        TRACE.append(dict(kind="conv", name=wkey, x=x, w=sd[wkey], scale=sc if bn is not None else None,
                          bias=bi if bn is not None else bias, stride=stride, padding=padding, res=res, relu=relu, y=y,
                          out_f32=out_f32))
    return y


def _lin_b(x, w, b, res=None, relu=False, out_f32=False):
    y = F.linear(x, _rb(w)) + b
    if res is not None:
        y = y + res
    if relu:
        y = F.relu(y)
    y = y if out_f32 else _rb(y)
    if TRACE is not None:
        TRACE.append(dict(kind="linear", name="linear", x=x, w=w, scale=None, bias=b, stride=1, padding=0, res=res, relu=relu,
                          y=y, out_f32=out_f32))
    return y


def _attention_b(q, k, v, nheads, tile=64):
    """q [B, Lq, E], k / v [B, Lk, E] (bf16-valued) -> [B, Lq, E]; online softmax over 64-key tiles as the kernel runs it"""
    B, Lq, E = q.shape
    Lk, D = k.shape[1], E // nheads
    qh = q.reshape(B, Lq, nheads, D).permute(0, 2, 1, 3)
    kh = k.reshape(B, Lk, nheads, D).permute(0, 2, 1, 3)
    vh = v.reshape(B, Lk, nheads, D).permute(0, 2, 1, 3)
    c = torch.tensor(0.1767766952966369, dtype=torch.float32) * torch.tensor(1.44269504088896341, dtype=torch.float32)
    m = torch.full((B, nheads, Lq, 1), -1e30)
    l = torch.zeros(B, nheads, Lq, 1)
    o = torch.zeros(B, nheads, Lq, D)
    for t in range(0, Lk, tile):
        s = qh @ kh[:, :, t:t + tile].transpose(-1, -2)
        m_new = torch.maximum(m, s.max(-1, keepdim=True).values)
        alpha = torch.exp2((m - m_new) * c)
        p = torch.exp2(s * c - m_new * c)
        l = l * alpha + p.sum(-1, keepdim=True)
        o = o * alpha + _rb(p) @ vh[:, :, t:t + tile]
        m = m_new
    return _rb(o * (1.0 / l)).permute(0, 2, 1, 3).reshape(B, Lq, E)


def _ln_b(x, sd, p, pos=None):
    y = F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    return _rb(y), (_rb(y + pos) if pos is not None else None)


@torch.no_grad()
def detr_forward_bf16(sd, img, blocks=(3, 4, 6, 3), nheads=8, enc_layers=6, dec_layers=6, return_taps=False):
    """detr_forward with bf16 storage where the gfx950 bf16 mode has it (see above); same output dict, heads in fp32."""
    pre = "backbone.0.body."
    x = _conv_b(_rb(img), sd, pre + "conv1.weight", bn=pre + "bn1", stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for l in range(4):
        for i in range(blocks[l]):
            p = f"{pre}layer{l + 1}.{i}."
            stride = 2 if (i == 0 and l > 0) else 1
            t = _conv_b(x, sd, p + "conv1.weight", bn=p + "bn1", relu=True)
            t = _conv_b(t, sd, p + "conv2.weight", bn=p + "bn2", stride=stride, padding=1, relu=True)
            idt = _conv_b(x, sd, p + "downsample.0.weight", bn=p + "downsample.1", stride=stride) if i == 0 else x
            x = _conv_b(t, sd, p + "conv3.weight", bn=p + "bn3", res=idt, relu=True)
    feat = x
    B, _, h, w = feat.shape
    E = sd["input_proj.weight"].shape[0]
    pos = position_embedding(h, w, batch=1).flatten(2).permute(0, 2, 1)            # [1, L, E] fp32 table
    src = _conv_b(feat, sd, "input_proj.weight", bias=sd["input_proj.bias"]).flatten(2).permute(0, 2, 1)   # [B, L, E]
    srcpos = _rb(src + pos)
    tp = "transformer."
    for i in range(enc_layers):
        p = f"{tp}encoder.layers.{i}."
        W, b = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
        qk = _lin_b(srcpos, W[:2 * E], b[:2 * E])
        v = _lin_b(src, W[2 * E:], b[2 * E:])
        att = _attention_b(qk[..., :E], qk[..., E:], v, nheads)
        tmp = _lin_b(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], res=src)
        src, _ = _ln_b(tmp, sd, p + "norm1")
        ffn = _lin_b(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"], relu=True)
        tmp = _lin_b(ffn, sd[p + "linear2.weight"], sd[p + "linear2.bias"], res=src)
        src, srcpos = _ln_b(tmp, sd, p + "norm2", pos)
    memory = src
    qpos = sd["query_embed.weight"].unsqueeze(0)                                   # [1, Q, E] fp32
    Q = qpos.shape[1]
    tgt = torch.zeros(B, Q, E)
    tgtpos = _rb(qpos).expand(B, Q, E)
    for i in range(dec_layers):
        p = f"{tp}decoder.layers.{i}."
        W, b = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
        qk = _lin_b(tgtpos, W[:2 * E], b[:2 * E])
        v = _lin_b(tgt, W[2 * E:], b[2 * E:])
        att = _attention_b(qk[..., :E], qk[..., E:], v, nheads)
        tmp = _lin_b(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], res=tgt)
        tgt, tgtpos = _ln_b(tmp, sd, p + "norm1", qpos)
        W, b = sd[p + "multihead_attn.in_proj_weight"], sd[p + "multihead_attn.in_proj_bias"]
        cq = _lin_b(tgtpos, W[:E], b[:E])
        kc = _lin_b(srcpos, W[E:2 * E], b[E:2 * E])
        vc = _lin_b(memory, W[2 * E:], b[2 * E:])
        att = _attention_b(cq, kc, vc, nheads)
        tmp = _lin_b(att, sd[p + "multihead_attn.out_proj.weight"], sd[p + "multihead_attn.out_proj.bias"], res=tgt)
        tgt, _ = _ln_b(tmp, sd, p + "norm2")
        ffn = _lin_b(tgt, sd[p + "linear1.weight"], sd[p + "linear1.bias"], relu=True)
        tmp = _lin_b(ffn, sd[p + "linear2.weight"], sd[p + "linear2.bias"], res=tgt)
        tgt, tgtpos = _ln_b(tmp, sd, p + "norm3", qpos)
    hs, _ = _ln_b(tgt, sd, tp + "decoder.norm")

    def mlp(p):
        h1 = _lin_b(hs, sd[p + ".layers.0.weight"], sd[p + ".layers.0.bias"], relu=True)
        h2 = _lin_b(h1, sd[p + ".layers.1.weight"], sd[p + ".layers.1.bias"], relu=True)
        return _lin_b(h2, sd[p + ".layers.2.weight"], sd[p + ".layers.2.bias"], out_f32=True)
    out = {
        "pred_logits": _lin_b(hs, sd["class_embed.weight"], sd["class_embed.bias"], out_f32=True),
        "pred_boxes": mlp("bbox_embed").sigmoid(),
        "pred_angle": mlp("angle_embed"),
        "pred_offset": mlp("offset_embed"),
        "pred_size": mlp("size_embed"),
        "pred_depth": mlp("depth_embed"),
        "pred_obj_features": hs,
    }
    if return_taps:
        out["_layer4"] = feat
        out["_memory"] = memory
    return out


# ---- post-processing (host) ---------------------------------------------------------------------------
def iou_2d(a, b):
    x_min = max(a[0, 0], b[0, 0]); y_min = max(a[0, 1], b[0, 1])
    x_max = min(a[1, 0], b[1, 0]); y_max = min(a[1, 1], b[1, 1])
    inter = max(0, (x_max - x_min)) * max(0, (y_max - y_min))
    return inter / (np.prod(a[1] - a[0]) + np.prod(b[1] - b[0]) - inter)


def iou_3d(a, b):
    lo = np.maximum(a[0], b[0]); hi = np.minimum(a[1], b[1])
    inter = max(0, hi[0] - lo[0]) * max(0, hi[1] - lo[1]) * max(0, hi[2] - lo[2])
    return inter / (np.prod(a[1] - a[0]) + np.prod(b[1] - b[0]) - inter)


def nms_3d(class_, scores, t_cos, dimensions, bbox_2d, nms_2d=True):
    orders = np.argsort(scores)[::-1]
    keep, suppressed = [], set()

    def aabb(i):
        d = dimensions[i]
        return np.array([[-d[0], -d[1], -d[2]], [d[0], d[1], d[2]]]) / 2. + t_cos[i][None, :]

    for i, s in enumerate(orders):
        if s in suppressed:
            continue
        keep.append(s)
        sb = aabb(s)
        for t in orders[i + 1:]:
            if t in suppressed:
                continue
            if class_[t] == class_[s] and iou_3d(sb, aabb(t)) > 0.25:
                suppressed.add(t)
                continue
            if nms_2d and iou_2d(bbox_2d[s], bbox_2d[t]) > 0.5:
                suppressed.add(t)
    return keep


def postprocess(out, img_size, threshold, intr_mat, nms_2d=True):
    img_w, img_h = img_size
    f = torch.tensor([intr_mat[0, 0], intr_mat[1, 1]]).float()
    c = torch.tensor([intr_mat[0, 2], intr_mat[1, 2]]).float()
    res = {k: [] for k in ("bboxes", "dimensions", "angles", "translates", "classes", "scores")}
    for i in range(len(out["pred_logits"])):
        n = len(out["pred_boxes"][i])
        probas = out["pred_logits"].softmax(-1)[i, :, :-1]
        keep = probas.max(-1).values > threshold
        score = probas.max(-1).values[keep].numpy()
        cls = probas.max(-1).indices[keep].numpy()
        xc, yc, w, h = out["pred_boxes"][i].unbind(-1)
        boxes = torch.stack([xc - 0.5 * w, yc - 0.5 * h, xc + 0.5 * w, yc + 0.5 * h], -1)
        rb = (boxes * torch.tensor([img_w, img_h, img_w, img_h]).float()).reshape(n, 2, 2)
        ro = out["pred_offset"][i] * torch.tensor([img_w, img_h]).float()
        centre = (ro + torch.mean(rb, dim=1) - c.unsqueeze(0)) / f.unsqueeze(0)
        centre = centre * out["pred_depth"][i]
        centre = torch.cat([centre, out["pred_depth"][i]], dim=1)
        nb = out["pred_angle"].shape[-1]
        angle = (out["pred_angle"][i].max(-1).indices * (180 / nb))[keep].numpy()
        dim = out["pred_size"][i, keep].numpy()
        rbk = rb[keep].numpy()
        t_co = centre[keep].numpy()
        k = nms_3d(cls, score, t_co, dim, rbk, nms_2d)
        res["dimensions"].append(dim[k]); res["bboxes"].append(rbk[k]); res["translates"].append(t_co[k])
        res["classes"].append(cls[k]); res["scores"].append(score[k]); res["angles"].append(angle[k])
    return res
