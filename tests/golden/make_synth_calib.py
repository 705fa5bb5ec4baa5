"""Data-dependent FrozenBN statistics for the synthetic detector weights (odam_amd/weights.py, calibrated=True).

Why: a 50/101-layer ReLU network with zero-mean random filters and identity batch-norm statistics maps every
input to nearly the same feature vector (all 850 memory tokens, hence all 100 decoder queries, come out equal): the
synthetic detector then yields ONE detection per frame however the heads are biased, and nothing downstream of the
detector (NMS, association, multi-object tracks, merge) sees realistic load.  A trained checkpoint's BN statistics
are the per-channel mean / variance of its own activations; this script gives the synthetic weights the same
property: layer by layer, running_mean / running_var := the statistics of the conv output on a few seeded synthetic
frames (torch CPU).  The result is stored (float16, exactly representable on every box) as
odam_amd/data/synth_calib_<backbone>_s<seed>.npz, so weights are identical wherever they are generated.
Second, the mean decoder output over queries and frames ("hs_mean") is stored: the prediction heads of the scene
weights are centred on it (odam_amd/weights.py), so that class / box / depth outputs follow the differences between
queries instead of their common component.  (Uses the CPU oracle's transformer, hence lives with the test tooling.)
Run: python tests/golden/make_synth_calib.py [resnet50|resnet101] [seed]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def calibrate(backbone="resnet50", seed=0, n_frames=6, h=256, w=320):
    from odam_amd import weights, synth
    sd = weights.make_state_dict(backbone=backbone, seed=seed)
    weights.scene_backbone_gains(sd)
    blocks = weights.RESNET_BLOCKS[backbone]
    frames = list(synth.make_frames(n_frames, h, w, seed=100 + seed))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = (torch.stack([torch.from_numpy(f).permute(2, 0, 1).float() / 255 for f in frames]) - mean) / std
    out = {}

    def bn(y, p):
        """set the statistics of BN `p` from its input y, then apply it (backbone.py:46-56)"""
        m = y.mean((0, 2, 3)).half().float(); v = y.var((0, 2, 3), unbiased=False).clamp_min(1e-3).half().float()
        out[p + ".running_mean"] = m.half().numpy(); out[p + ".running_var"] = v.half().numpy()
        sc = sd[p + ".weight"] * (v + 1e-5).rsqrt()
        return y * sc.view(1, -1, 1, 1) + (sd[p + ".bias"] - m * sc).view(1, -1, 1, 1)

    with torch.no_grad():
        bb = "backbone.0.body."
        x = F.relu(bn(F.conv2d(x, sd[bb + "conv1.weight"], None, 2, 3), bb + "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        for l, nb in enumerate(blocks):
            for i in range(nb):
                p = f"{bb}layer{l + 1}.{i}."
                s = 2 if (i == 0 and l > 0) else 1
                o = F.relu(bn(F.conv2d(x, sd[p + "conv1.weight"]), p + "bn1"))
                o = F.relu(bn(F.conv2d(o, sd[p + "conv2.weight"], None, s, 1), p + "bn2"))
                o = bn(F.conv2d(o, sd[p + "conv3.weight"]), p + "bn3")
                idt = bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, s), p + "downsample.1") if i == 0 else x
                x = F.relu(o + idt)
        # decoder output statistics with the calibrated backbone and the scene transformer
        import detr_oracle as O
        full = dict(sd)
        for k, v in out.items():
            full[k] = torch.from_numpy(v.astype(np.float32))
        weights.scene_transformer(full)
        B, _, fh, fw = x.shape
        src = F.conv2d(x, full["input_proj.weight"], full["input_proj.bias"])
        hs, _ = O.transformer(src, O.position_embedding(fh, fw, batch=B), full["query_embed.weight"], full)
        out["hs_mean"] = hs[-1].mean((0, 1)).half().numpy()
        out["hs_std"] = np.float16(hs[-1].std(1).mean().item())
    return out


if __name__ == "__main__":
    bbn = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.set_num_threads(4)
    res = calibrate(bbn, seed)
    path = os.path.join(REPO, "odam_amd", "data", f"synth_calib_{bbn}_s{seed}.npz")
    np.savez_compressed(path, **res)
    print(path, os.path.getsize(path) // 1024, "KiB;", (len(res) - 2) // 2, "batch-norm layers; query spread", float(res["hs_std"]))
