"""Image transforms of the detector input: the reference's `get_transforms()` (likojack/ODAM
src/datasets/transforms.py:281-290): shortest side -> 800 (max 1333) with PIL bilinear on the uint8 image,
ToTensor (/255), Normalize(ImageNet mean/std).  Caller-supplied object as in the reference
(`transforms(rgb, None) -> (tensor[3,H,W], None)`), on the host with PIL + torch (torchvision is not required), or --
`get_device_transforms()` -- on the device through `odam_detr_preprocess_u8`."""
import numpy as np
import torch
from PIL import Image

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def target_size(w, h, size=800, max_size=1333):
    """transforms.py:78-96 get_size_with_aspect_ratio -> (oh, ow)"""
    mn, mx = float(min(w, h)), float(max(w, h))
    if mx / mn * size > max_size:
        size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


class Transforms:
    def __init__(self, size=800, max_size=1333):
        self.size, self.max_size = size, max_size
        self.mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)

    def __call__(self, image, target=None):
        if isinstance(image, np.ndarray):
            image = Image.fromarray(image)
        oh, ow = target_size(image.size[0], image.size[1], self.size, self.max_size)
        img = image.resize((ow, oh), Image.BILINEAR)  # torchvision F.resize on a PIL image
        x = torch.from_numpy(np.asarray(img, np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div(255)  # to_tensor
        x = (x - self.mean) / self.std  # F.normalize: tensor.sub_(mean).div_(std)
        return x, target


def get_transforms():
    return Transforms()


class DeviceTransforms(Transforms):
    """Same transform, evaluated by the library on the device (`Detector.preprocess_u8`, bit-identical results):
    pass it to OdamProcess instead of get_transforms() and `detect_frames` uploads the raw uint8 frames (11x fewer
    PCIe bytes) and resizes / normalises them on the GPU.  Called directly it is the host transform."""
    on_device = True


def get_device_transforms():
    return DeviceTransforms()
