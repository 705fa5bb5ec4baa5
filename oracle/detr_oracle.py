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


def resnet_body(x, sd, blocks=(3, 4, 6, 3), prefix="backbone.0.body."):
    x = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, prefix + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for l in range(4):
        for i in range(blocks[l]):
            p = f"{prefix}layer{l + 1}.{i}."
            stride = 2 if (i == 0 and l > 0) else 1
            idt = x
            out = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
            out = F.relu(frozen_bn(F.conv2d(out, sd[p + "conv2.weight"], None, stride=stride, padding=1), sd, p + "bn2"))
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


def transformer(src, pos, query_embed, sd, nheads=8, enc_layers=6, dec_layers=6, prefix="transformer."):
    """src [B,C,h,w], pos [B,C,h,w], query_embed [Q,C] -> hs [dec_layers,B,Q,C], memory [HW,B,C]"""
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
def detr_forward(sd, img, blocks=(3, 4, 6, 3), nheads=8, enc_layers=6, dec_layers=6, return_taps=False):
    """img [B,3,H,W] float32 CPU -> the reference's output dict for the last decoder layer."""
    feat = resnet_body(img, sd, blocks)
    B, _, h, w = feat.shape
    pos = position_embedding(h, w, batch=B)
    src = F.conv2d(feat, sd["input_proj.weight"], sd["input_proj.bias"])
    hs, memory = transformer(src, pos, sd["query_embed.weight"], sd, nheads, enc_layers, dec_layers)
    out = heads(hs[-1], sd)
    if return_taps:
        out["_layer4"] = feat
        out["_memory"] = memory.permute(1, 0, 2)  # [B, HW, C]
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
