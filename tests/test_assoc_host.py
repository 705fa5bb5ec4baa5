"""odam_assoc_attach (the host-side Hungarian step + score tests of OdamProcess's fast path, include/odam_assoc.h) against what it
restates: scipy.optimize.linear_sum_assignment(1 - scores) as hungarian_matching calls it (src/models/associator.py:19-35) and the
comparisons of _attach_to_tracks (src/processor.py:228-231).  A host function: runs without a GPU."""
import ctypes
import os

import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from odam_amd import _lib

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _want(score_mat, mt, st):
    """_attach_fast as it was written in numpy (round 6, before the native call)"""
    n_det = score_mat.shape[1] - 1
    scores = score_mat[:-1, :-1]
    r_, c_ = linear_sum_assignment(1 - scores)
    ok = scores[r_, c_] > mt
    match = np.full(n_det, -1, np.int32)
    match[c_[ok]] = r_[ok]
    keep = ~(score_mat[match, np.arange(n_det)] < st)
    return match, keep


def _got(score_mat, mt, st):
    sm = np.ascontiguousarray(score_mat, np.float32)
    n_det = sm.shape[1] - 1
    match = np.full(n_det, -7, np.int32)
    keep = np.full(n_det, 9, np.uint8)
    rc = _lib.lib().odam_assoc_attach(sm.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(sm.shape[0] - 1), ctypes.c_int(n_det), ctypes.c_int(sm.shape[1]),
                                      ctypes.c_double(mt), ctypes.c_double(st), match.ctypes.data_as(ctypes.c_void_p), keep.ctypes.data_as(ctypes.c_void_p))
    return rc, match, keep.astype(bool)


def test_attach_on_the_host_equals_scipy_and_numpy():
    rng = np.random.default_rng(11)
    cases = []
    for T, n in ((5, 3), (3, 5), (30, 30), (42, 17), (17, 30), (100, 30), (128, 30), (1, 1), (1, 30), (60, 1), (31, 2), (2, 31), (0, 4), (4, 0)):
        for rep in range(3):
            cases.append(rng.random((T + 1, n + 1), dtype=np.float32))
    for T, n in ((6, 6), (9, 4), (4, 9), (40, 30), (30, 12)):
        cases.append(np.full((T + 1, n + 1), 0.5, np.float32))                                   # scipy: the identity on a constant matrix
        cases.append((rng.integers(0, 3, (T + 1, n + 1)) / 4).astype(np.float32))                # three distinct values: exact ties everywhere
        cases.append((rng.integers(0, 8, (T + 1, n + 1)) / 8).astype(np.float32))
        a = rng.random((T + 1, n + 1), dtype=np.float32); a[1::2] = a[0::2][:len(a[1::2])]      # duplicated rows
        cases.append(a)
        b = rng.random((T + 1, n + 1), dtype=np.float32); b[:, 1::2] = b[:, 0::2][:, :b[:, 1::2].shape[1]]      # duplicated columns
        cases.append(b)
    # thresholds that are not float32 numbers and scores exactly at their float32 neighbours: the comparisons are float32 comparisons
    edge = rng.random((8, 6), dtype=np.float32)
    edge[0, 0] = np.float32(0.2); edge[1, 1] = np.nextafter(np.float32(0.2), np.float32(1)); edge[7, 2] = np.float32(0.8); edge[2, 3] = np.float32(0.8)
    cases.append(edge)
    n = 0
    for mt, st in ((0.1, 0.8), (0.2, 0.8), (0.0, 0.2), (0.6, 0.6)):
        for sm in cases:
            rc, match, keep = _got(sm, mt, st)
            assert rc == 0
            if sm.shape[0] > 1 and sm.shape[1] > 1:
                wm, wk = _want(sm, mt, st)
            else:      # an empty side: nothing is matched; every detection reads the dustbin row
                wm = np.full(sm.shape[1] - 1, -1, np.int32)
                wk = ~(sm[-1, :-1] < st)
            assert np.array_equal(match, wm) and np.array_equal(keep, wk), (sm.shape, mt, st, match, wm)
            n += 1
    assert n >= 250


def test_attach_on_the_reference_run_assignments():
    """every log-assignment matrix of the reference-run fixtures, through torch's exp as the fast path computes it"""
    n_fix = 0
    for name in ("e2e.npz", "process_tracks_net.npz"):
        z = np.load(os.path.join(GOLD, name), allow_pickle=True)
        for k in [k for k in z.files if k.startswith("Z") and k[1:].isdigit()]:
            Z = torch.from_numpy(z[k].astype(np.float32))
            if Z.shape[0] < 2 or Z.shape[1] < 2:
                continue
            sm = Z.exp().numpy()
            for mt, st in ((0.1, 0.8), (0.2, 0.5)):
                rc, match, keep = _got(sm, mt, st)
                wm, wk = _want(sm, mt, st)
                assert rc == 0 and np.array_equal(match, wm) and np.array_equal(keep, wk), (name, k, mt)
                n_fix += 1
    assert n_fix >= 100


def test_attach_reports_what_scipy_raises_for():
    sm = np.random.default_rng(0).random((5, 4), dtype=np.float32)
    bad = sm.copy(); bad[1, 2] = np.nan
    assert _got(bad, 0.1, 0.8)[0] == 4
    with pytest.raises(ValueError):
        linear_sum_assignment(1 - bad[:-1, :-1])
    inf = sm.copy(); inf[:-1, :-1] = -np.inf          # 1 - score = +inf everywhere: infeasible
    assert _got(inf, 0.1, 0.8)[0] == 5
    with pytest.raises(ValueError):
        linear_sum_assignment(1 - inf[:-1, :-1])
