"""Golden for the configuration variants the shipped yaml does not use: `pre_norm: True` (src/models/transformer.py:169-188,
240-262, 26-28) `position_embedding: 'learned'` (src/models/position_encoding.py:49-74, 83-84) and `dilation: True` (src/models/backbone.py:89-91;
the dilated layer4 itself is torchvision's, restated in make_golden_detr._ResNet -- the body stays "parity unpinned").  The reference's DETR is
built from its own config with those two keys changed and run AS IT IS on the CPU (ResNet body: the stand-in of
make_golden_detr.py, torchvision not installed), round-1 weights seed 0 + odam_amd.weights.add_variant_weights.
Writes tests/golden/detr_variants.npz (2 frames of 3x256x320 regenerated from the seed; three model variants).
Run: python tests/golden/make_golden_detr_variants.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

VARIANTS = {"pre": dict(pre_norm=True, position_embedding="sine"), "learned": dict(pre_norm=False, position_embedding="learned"),
            "pre_learned": dict(pre_norm=True, position_embedding="learned"), "dc5": dict(dilation=True)}
KEYS = ("pred_logits", "pred_boxes", "pred_angle", "pred_offset", "pred_size", "pred_depth")


def image():
    torch.manual_seed(321)
    return torch.randn(2, 3, 256, 320)


def main():
    import refenv
    refenv.setup()
    import torchvision
    from make_golden_detr import _ResNet
    torchvision.models.resnet50 = lambda replace_stride_with_dilation=None, pretrained=False, norm_layer=None: _ResNet((3, 4, 6, 3), norm_layer, replace_stride_with_dilation)
    from src.config.configs import ConfigLoader
    from src.models.detr import build as build_detector
    from odam_amd import weights
    sd = weights.add_variant_weights(weights.make_state_dict(seed=0))
    data = {}
    for name, kv in VARIANTS.items():
        cfg = ConfigLoader().merge_cfg(["/root/reference/configs/detr_scan_net.yaml"])
        cfg.device = "cpu"
        for k, v in kv.items():
            setattr(cfg, k, v)
        model, _, _ = build_detector(cfg)
        res = model.load_state_dict(sd, strict=False)
        assert all(k.startswith("backbone.0.body.fc") for k in res.missing_keys), res.missing_keys      # every tensor the variant reads was supplied
        model.eval()
        with torch.no_grad():
            out = model(image())
        for k in KEYS:
            data[f"{name}_{k}"] = out[k].numpy()
    np.savez_compressed(os.path.join(HERE, "detr_variants.npz"), **data)
    base = np.load(os.path.join(HERE, "detr_small.npz"))
    print("detr_variants.npz:", {k: v.shape for k, v in data.items() if k.endswith("logits")})


if __name__ == "__main__":
    main()
