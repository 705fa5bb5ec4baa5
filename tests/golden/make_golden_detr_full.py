"""Full-size golden for the detector: the reference's DETR (likojack/ODAM src/models/detr.py, transformer.py,
backbone.py Joiner / FrozenBatchNorm2d / position encoding, with the stand-in ResNet body of make_golden_detr.py)
run on the CPU on ONE 3x800x1066 frame -- the size every ScanNet frame is resized to, 850 encoder tokens.
Writes tests/golden/detr_full.npz (input seed + the six output tensors).   Run: python tests/golden/make_golden_detr_full.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden_detr import _ResNet  # noqa: E402


def main():
    import refenv
    refenv.setup()
    import torchvision
    torchvision.models.resnet50 = lambda replace_stride_with_dilation=None, pretrained=False, norm_layer=None: _ResNet((3, 4, 6, 3), norm_layer)
    from src.config.configs import ConfigLoader
    from src.models.detr import build as build_detector
    from odam_amd import weights

    cfg = ConfigLoader().merge_cfg(["/root/reference/configs/detr_scan_net.yaml"])
    cfg.device = "cpu"
    model, _, _ = build_detector(cfg)
    sd = weights.make_state_dict(seed=0)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("backbone.0.body.fc") for k in missing.missing_keys), missing
    model.eval()
    torch.manual_seed(321)
    img = torch.randn(1, 3, 800, 1066)
    with torch.no_grad():
        out = model(img)
    data = {"img_seed": np.int32(321)}
    for k in ("pred_logits", "pred_boxes", "pred_angle", "pred_offset", "pred_size", "pred_depth"):
        data[k] = out[k].numpy()
    np.savez_compressed(os.path.join(HERE, "detr_full.npz"), **data)
    print("detr_full.npz written:", {k: v.shape for k, v in data.items() if k != "img_seed"})


if __name__ == "__main__":
    main()
