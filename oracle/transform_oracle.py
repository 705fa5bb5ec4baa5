"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the detector's input transform:
    likojack/ODAM src/datasets/transforms.py:75-105  resize -> torchvision F.resize(PIL image) -> PIL Image.resize(BILINEAR)
    transforms.py:222-224  ToTensor   -> uint8 -> float32, / 255
    transforms.py:236-243  Normalize  -> (x - mean) / std in float32

The arithmetic of the resize lives in a third-party dependency, Pillow (src/libImaging/Resample.c; the reference
pins pillow 8.4.0 in environment.yml, this image has 12.2.0 -- the 8-bit resampling code is the same in both):
two passes (horizontal, then vertical), each output byte = clip8((2^21 + sum_k px[xmin + k] * K[k]) >> 22) with
K[k] = (int)(w_k * 2^22 +- 0.5), w_k the normalised triangle-filter weights of `precompute_coeffs`.
Pinned against PIL itself in tests/test_transform.py (bit-exact on random and structured images)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter and the
    full-image box: (xmin[out], count[out], K[out, ksize] int32)"""
    scale = in_size / out_size            # double division, as in C
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    K = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(ksize)
        ww = 0.0
        for x in range(n):
            t = (x + lo - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        for x in range(n):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            K[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)   # C cast: truncation toward zero
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, K


def _pass(img, xmin, cnt, K, axis):
    """one resampling pass over `axis` of a uint8 [H, W, C] image"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((len(xmin),) + src.shape[1:], np.uint8)
    for xx in range(len(xmin)):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(cnt[xx]):
            acc += src[xmin[xx] + k] * int(K[xx, k])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, ow, oh):
    """PIL Image.resize((ow, oh), BILINEAR) of a uint8 [h, w, 3] array: horizontal pass, then vertical"""
    h, w = img.shape[:2]
    out = img
    if ow != w:
        out = _pass(out, *coeffs(w, ow), axis=1)
    if oh != h:
        out = _pass(out, *coeffs(h, oh), axis=0)
    return out


def to_tensor_normalize(u8):
    """[H, W, 3] uint8 -> float32 [3, H, W]: ToTensor then Normalize, float32 arithmetic"""
    x = u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    mean = np.asarray(MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(STD, np.float32).reshape(3, 1, 1)
    return (x - mean) / std


def transform(img, oh, ow):
    return to_tensor_normalize(resize_bilinear_u8(img, ow, oh))
