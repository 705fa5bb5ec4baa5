"""Input transform: the oracle's restatement of Pillow's 8-bit bilinear resampling against
Pillow itself, and against the host transform the reference's get_transforms() amounts to (CPU only)."""
import os
import sys

import numpy as np
import pytest
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import transform_oracle as T  # noqa: E402


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 1066, 800), (800, 600, 1066, 800), (64, 48, 107, 80), (33, 57, 47, 80),
                                       (100, 80, 50, 40), (97, 31, 97, 62), (31, 97, 64, 97), (5, 4, 3, 9)])
def test_resize_oracle_is_bit_exact_vs_pillow(w, h, ow, oh):
    rng = np.random.default_rng(w * 7 + h)
    for kind in range(3):
        if kind == 0:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif kind == 1:
            img = ((np.add.outer(np.arange(h), np.arange(w))[..., None] * np.array([3, 5, 7])) % 256).astype(np.uint8)
        else:
            img = np.zeros((h, w, 3), np.uint8); img[::2, 1::2] = 255
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(T.resize_bilinear_u8(img, ow, oh), ref), kind


def test_full_transform_oracle_equals_host_transform():
    from odam_amd import transforms
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    x, _ = transforms.Transforms()(img)
    assert tuple(x.shape) == (3, 800, 1066)
    assert np.array_equal(x.numpy(), T.transform(img, 800, 1066))


def test_coefficients_are_normalised_and_local():
    for n_in, n_out in ((480, 800), (640, 1066), (800, 600), (7, 7)):
        xmin, cnt, K = T.coeffs(n_in, n_out)
        assert (cnt >= 1).all() and (xmin >= 0).all() and (xmin + cnt <= n_in).all()
        s = np.array([K[i, :cnt[i]].sum() for i in range(n_out)])
        assert np.abs(s - (1 << 22)).max() <= 2        # weights sum to one in 22-bit fixed point (rounding)
