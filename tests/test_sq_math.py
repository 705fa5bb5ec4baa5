"""odam_amd/csrc/sq_math.h (the device math of the SQ kernel), built for the host, against libm."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sqmath") / "libsqmathcheck.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", out,
                           os.path.join(REPO, "tests", "native", "sq_math_check.cpp"), "-lm"])
    L = ctypes.CDLL(out)
    L.sq_math_sweep1.restype = ctypes.c_long
    L.sq_math_sweep_pow.restype = ctypes.c_long
    return L


def bits(f):
    return int(np.float32(f).view(np.uint32))


@pytest.mark.parametrize("which,name,hi", [(0, "sinf", np.pi), (1, "cosf", np.pi), (3, "logf", 3e38)])
def test_bit_identical_to_libm(chk, which, name, hi):
    fb = ctypes.c_uint32()
    # every 7th float of the whole range, both signs (the exhaustive sweep was run once: 0 mismatches)
    n = chk.sq_math_sweep1(which, ctypes.c_uint32(0), ctypes.c_uint32(bits(hi)), ctypes.c_uint32(7), ctypes.byref(fb))
    assert n == 0, f"{name}: {n} mismatches, e.g. bits {hex(fb.value)}"


def test_expf_bit_identical_on_sigmoid_range(chk):
    fb = ctypes.c_uint32()
    n = chk.sq_math_sweep1(2, ctypes.c_uint32(0), ctypes.c_uint32(bits(40.0)), ctypes.c_uint32(5), ctypes.byref(fb))
    assert n == 0, hex(fb.value)


def test_powf_bit_identical_on_sampler_domain(chk):
    ys = np.concatenate([np.linspace(0.2, 1.6, 29), np.random.RandomState(0).uniform(0.2, 1.6, 35)]).astype(np.float32)
    n = chk.sq_math_sweep_pow(ctypes.c_uint32(bits(1e-10)), ctypes.c_uint32(bits(1.0)), ctypes.c_uint32(1009),
                              ys.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(ys))
    assert n == 0
