"""Golden for mixed-size batches: the reference's DETR.forward on a LIST of three images of different sizes
(src/models/detr.py:64-65 -> nested_tensor_from_tensor_list, src/utils/misc.py:303-320: zero padding + boolean mask;
mask reduced to the token grid in backbone.py:79; per-image sine embedding position_encoding.py:26-46;
key_padding_mask in transformer.py:157-160 and :224-228), scene weights, CPU.  ResNet body: the stand-in of
make_golden_detr.py (torchvision not installed).  Writes tests/golden/detr_mixed.npz (inputs are regenerated from
the seed).  Run: python tests/golden/make_golden_detr_mixed.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SIZES = [(224, 320), (256, 288), (192, 256)]      # (H, W): padded to 256 x 320 -> 8 x 10 tokens


def images():
    g = torch.Generator().manual_seed(77)
    return [torch.randn(3, h, w, generator=g) for h, w in SIZES]


def main():
    import refenv
    refenv.setup()
    import torchvision
    from make_golden_detr import _ResNet
    torchvision.models.resnet50 = lambda replace_stride_with_dilation=None, pretrained=False, norm_layer=None: _ResNet((3, 4, 6, 3), norm_layer)
    from src.config.configs import ConfigLoader
    from src.models.detr import build as build_detector
    from odam_amd import weights
    cfg = ConfigLoader().merge_cfg(["/root/reference/configs/detr_scan_net.yaml"])
    cfg.device = "cpu"
    model, _, _ = build_detector(cfg)
    model.load_state_dict(weights.make_state_dict(seed=0, scene=True), strict=False)
    model.eval()
    with torch.no_grad():
        out = model(images())
        # for contrast: the first image alone (no padding) -- padding changes its result, i.e. the mask path matters
        alone = model(images()[0][None])
    d = {k: out[k].numpy() for k in ("pred_logits", "pred_boxes", "pred_angle", "pred_offset", "pred_size", "pred_depth")}
    d["alone_logits0"] = alone["pred_logits"].numpy()
    d["sizes"] = np.asarray(SIZES, np.int32)
    np.savez_compressed(os.path.join(HERE, "detr_mixed.npz"), **d)
    print("detr_mixed.npz:", {k: v.shape for k, v in d.items()}, "padded-vs-alone logits diff %.3f" % np.abs(d["pred_logits"][0] - d["alone_logits0"][0]).max())


if __name__ == "__main__":
    main()
