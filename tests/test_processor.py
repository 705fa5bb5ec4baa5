"""OdamProcess (plugin surface) driven exactly like the reference's was when tests/golden/process_tracks.npz
was made: canned detections per frame, the same deterministic stand-in associator."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)


class _OraclePoints:
    """fitter stand-in for the CPU run: surface points from the oracle (tests only)"""

    def __init__(self, oracle):
        self.o = oracle

    def points(self, params):
        import torch
        return torch.from_numpy(np.stack([self.o.points(p) for p in np.asarray(params, np.float32).reshape(-1, 9)]))


def _drive(fitter, z):
    from make_golden_process import FakeAssociator
    from odam_amd.processor import OdamProcess

    class _Det:           # only `.device` is read by process_frame
        device = "cpu"
    proc = OdamProcess(_Det(), FakeAssociator(), None, None, fitter=fitter)
    proc.init_sequence(z["K"], 480, 640)
    for i, fid in enumerate(z["img_names"]):
        rows = [list(r) for r in z[f"det{int(fid)}"]]
        proc.process_frame(None, int(fid), z["scene_T_wcs"][i], detections=rows)
    return proc


def _check(proc, z):
    assert len(proc.tracks) == int(z["n_tracks"])
    assert np.array_equal(np.asarray(proc.usable_frames), z["usable_frames"])
    assert np.array_equal(np.asarray(proc.T_wcs), z["T_wcs"])
    assert np.array_equal(np.asarray(proc.P_cws), z["P_cws"])
    for i, t in enumerate(proc.tracks):
        ref = z[f"track{i}"]
        assert t.shape == ref.shape
        assert np.array_equal(t[:, :2], ref[:, :2])          # frame ids and class labels: exact
        assert np.allclose(t, ref, rtol=1e-9, atol=1e-9)      # float64 host arithmetic


def test_process_frame_sequence_matches_reference(oracle, golden):
    z = golden("process_tracks.npz")
    _check(_drive(_OraclePoints(oracle), z), z)


@pytest.mark.gpu
def test_process_frame_sequence_matches_reference_gpu(golden):
    from odam_amd import sq
    z = golden("process_tracks.npz")
    _check(_drive(sq.SqFitter("cuda:0", 1), z), z)
