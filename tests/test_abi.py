"""The C-ABI shared library: loads without a GPU, exports every entry point include/*.h declares,
reports errors by return code.  No device compute here."""
import ctypes
import glob
import os
import re

import numpy as np

from conftest import REPO


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b((?:odam_[a-z0-9_]+)|sample_on_batch)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from odam_amd import _lib
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 6 and "sample_on_batch" in syms
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ but not exported by libodam_amd.so"


def test_error_convention():
    from odam_amd import _lib
    L = _lib.lib()
    rc = L.odam_sq_sample(None, None, None, None)
    assert rc == 1  # ODAM_E_INVALID
    assert b"odam_sq_sample" in L.odam_last_error()


def test_host_sampler_bit_exact_vs_oracle(oracle):
    """odam_sq_sample is the product's host restatement of sample_on_batch (same leaf code as the kernel)."""
    from odam_amd import sq
    rs = np.random.RandomState(11)
    for _ in range(200):
        a = rs.uniform(0.02, 1.6, 3).astype(np.float32); e = rs.uniform(0.2, 1.6, 2).astype(np.float32)
        et, om = sq.host_sample(a, e)
        oe, oo = oracle.sample(a, e)
        assert np.array_equal(et.view(np.uint32), oe.view(np.uint32))
        assert np.array_equal(om.view(np.uint32), oo.view(np.uint32))


def test_host_sampler_matches_reference_golden(golden):
    from odam_amd import sq
    z = golden("sq_sampler.npz")
    for a, e, et, om in zip(z["a"], z["e"], z["etas"], z["omegas"]):
        he, ho = sq.host_sample(a, e)
        assert np.array_equal(he.view(np.uint32), et.view(np.uint32))
        assert np.array_equal(ho.view(np.uint32), om.view(np.uint32))


def _ref_sampler():
    import pytest
    so = os.path.join(REPO, "oracle", "_ref", "libodam_ref_sampler.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference: make -C oracle ref)")
    return ctypes.CDLL(so)


def _call_sample_on_batch(L, shapes, eps, N, buffer_size, seed):
    fp = ctypes.POINTER(ctypes.c_float)
    B, M = shapes.shape[:2]
    et = np.full((B, M, N), np.nan, np.float32); om = np.full((B, M, N), np.nan, np.float32)
    L.sample_on_batch.restype = None
    L.sample_on_batch.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 5
    L.sample_on_batch(shapes.ctypes.data_as(fp), eps.ctypes.data_as(fp), et.ctypes.data_as(fp), om.ctypes.data_as(fp),
                      B, M, N, buffer_size, seed)
    return et, om


def test_sample_on_batch_has_the_reference_signature_and_bits():
    """`sample_on_batch(float*, float*, float*, float*, int B, int M, int N, int buffer_size, int seed)` exported by
    libodam_amd.so (what _sampler.pyx:430-439 binds) against the reference's own sampling.cpp compiled as it lies
    (oracle/_ref): B, M > 1, N != 1000, other buffer sizes and seeds -- bit for bit."""
    from odam_amd import _lib
    ref = _ref_sampler()
    L = _lib.lib()
    rs = np.random.RandomState(5)
    for B, M, N, nbuf, seed in ((1, 1, 1000, 201, 0), (2, 3, 257, 201, 0), (3, 2, 1000, 101, 7), (1, 5, 64, 513, 123456),
                                (4, 1, 1, 2, 1), (1, 2, 2048, 3, 99), (2, 2, 100, 1001, 2 ** 31 - 1)):
        shapes = rs.uniform(0.02, 1.6, (B, M, 3)).astype(np.float32)
        eps = rs.uniform(0.2, 1.6, (B, M, 2)).astype(np.float32)
        ge, go = _call_sample_on_batch(L, shapes.copy(), eps.copy(), N, nbuf, seed)
        re_, ro = _call_sample_on_batch(ref, shapes.copy(), eps.copy(), N, nbuf, seed)
        assert np.array_equal(ge.view(np.uint32), re_.view(np.uint32)), (B, M, N, nbuf, seed)
        assert np.array_equal(go.view(np.uint32), ro.view(np.uint32)), (B, M, N, nbuf, seed)


def test_sample_on_batch_agrees_with_the_pipeline_call(golden):
    """the reference binding's call (B = M = 1, N = 1000, buffer 201, seed 0) through the general entry point equals
    the Cython binding's golden output; a second primitive in the same batch continues the same random stream"""
    from odam_amd import _lib
    L = _lib.lib()
    z = golden("sq_sampler.npz")
    a, e = z["a"], z["e"]
    for k in range(len(a)):
        ge, go = _call_sample_on_batch(L, a[k][None, None].copy(), e[k][None, None].copy(), 1000, 201, 0)
        assert np.array_equal(ge[0, 0].view(np.uint32), z["etas"][k].view(np.uint32))
        assert np.array_equal(go[0, 0].view(np.uint32), z["omegas"][k].view(np.uint32))
    ge, go = _call_sample_on_batch(L, a[:2][None].copy(), e[:2][None].copy(), 1000, 201, 0)
    assert np.array_equal(ge[0, 0], z["etas"][0]) and not np.array_equal(ge[0, 1], z["etas"][1])
