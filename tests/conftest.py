import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_oracle():
    so = os.path.join(REPO, "oracle", "libodam_oracle.so")
    src = os.path.join(REPO, "oracle", "sq_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libodam_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


class Oracle:
    """ctypes view of oracle/libodam_oracle.so (the CPU checker; tests only)."""

    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)

    def __init__(self):
        self.lib = ctypes.CDLL(_ensure_oracle())
        z = np.load(os.path.join(REPO, "odam_amd", "data", "scale_prior.npz"))
        self.icov = z["icov"].astype(np.float32).reshape(8, 9)

    @staticmethod
    def P(x):
        return x.ctypes.data_as(Oracle.fp)

    def sample(self, a, e):
        a = np.ascontiguousarray(a, np.float32); e = np.ascontiguousarray(e, np.float32)
        et = np.zeros(1000, np.float32); om = np.zeros(1000, np.float32)
        self.lib.oracle_sq_sample(self.P(a), self.P(e), self.P(et), self.P(om))
        return et, om

    def points(self, p):
        p = np.ascontiguousarray(p, np.float32)
        out = np.zeros((1000, 3), np.float32)
        self.lib.oracle_sq_points(self.P(p), self.P(out))
        return out

    def loss_grad(self, p, P, tgt, mask, cls, s0, optimise_shapes=1):
        p = np.ascontiguousarray(p, np.float32)
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 12)
        tgt = np.ascontiguousarray(tgt, np.float32); mask = np.ascontiguousarray(mask, np.float32)
        F = len(tgt)
        g = np.zeros(9, np.float32); ext = np.zeros((F, 4), np.float32); win = np.zeros((F, 4), np.int32)
        l2d = ctypes.c_float(); l = ctypes.c_float()
        icov = self.P(self.icov[cls].copy()) if cls is not None and cls >= 0 else None
        s0 = np.ascontiguousarray(s0, np.float32)
        self.lib.oracle_sq_loss_grad(self.P(p), F, self.P(P), self.P(tgt), self.P(mask), icov, self.P(s0),
                                     optimise_shapes, ctypes.byref(l2d), ctypes.byref(l), self.P(g), self.P(ext),
                                     win.ctypes.data_as(self.ip))
        return l2d.value, l.value, g, ext, win

    def adam(self, p, m, v, g, t, n_opt=9):
        p, m, v = p.copy(), m.copy(), v.copy()
        g = np.ascontiguousarray(g, np.float32)
        self.lib.oracle_adam_step(self.P(p), self.P(m), self.P(v), self.P(g), t, n_opt, ctypes.c_double(0.01),
                                  ctypes.c_double(0.1))
        return p, m, v

    def fit(self, p0, P, tgt, mask, cls, n_iters=200, representation=0, want_traj=False, want_loss=False):
        p0 = np.ascontiguousarray(p0, np.float32)
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 12)
        tgt = np.ascontiguousarray(tgt, np.float32); mask = np.ascontiguousarray(mask, np.float32)
        out = np.zeros(9, np.float32)
        traj = np.zeros((n_iters, 9), np.float32) if want_traj else None
        loss = np.zeros(n_iters, np.float32) if want_loss else None
        icov = self.P(self.icov[cls].copy()) if cls is not None and cls >= 0 else None
        self.lib.oracle_sq_fit(self.P(p0), len(tgt), self.P(P), self.P(tgt), self.P(mask), icov, representation,
                               n_iters, self.P(out), self.P(traj) if want_traj else None,
                               self.P(loss) if want_loss else None)
        return out, traj, loss


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    return lambda name: np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b, floor=1e-3):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


# ---- what the tolerances were measured at: every test may record name -> value; at the end of a GPU session the table is
# written to gpurun_out/test_measured.json (scratch; a copy of a run is kept under profiles/), so that a tolerance in a
# test can be checked against what the kernels actually deliver
_MEASURED = {}


def record(name, value):
    """keep the largest value seen under a name"""
    v = float(value)
    if name not in _MEASURED or v > _MEASURED[name]:
        _MEASURED[name] = v


@pytest.fixture(scope="session")
def measured():
    return record


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "test_measured.json"), "w") as f:
            json.dump(dict(sorted(_MEASURED.items())), f, indent=1)
    except OSError:
        pass
