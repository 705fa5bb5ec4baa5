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


def test_preprocess_tracks_window_and_vectorisation():
    """`_preprocess_tracks` (all tracks at once, only the last 100 observations) against the per-track loop of the
    reference (processor.py:139-170 restated) on random tracks of 1..300 observations: identical float32 tensors."""
    from odam_amd.processor import OdamProcess, get_homogeneous, get_cam_azi
    rs = np.random.RandomState(4)
    proc = OdamProcess(None, None, None, None)
    proc.init_sequence(np.array([[577.87, 0, 319.5], [0, 577.87, 239.5], [0, 0, 1.0]]), 480, 640)
    A = rs.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
    T_wc = np.eye(4); T_wc[:3, :3] = Q; T_wc[:3, 3] = rs.normal(size=3)
    cam_azi = get_cam_azi(T_wc)
    tracks = []
    for n in (1, 2, 99, 100, 101, 300, 37):
        t = rs.normal(0, 1, (n, 82)); t[:, 14:78] = -1
        t[-1, -4:] = rs.uniform(-200, 900, 4)
        tracks.append(t)
    got = proc._preprocess_tracks([t.copy() for t in tracks], T_wc, cam_azi)
    want = np.full((len(tracks), 100, 79), -1.0, np.float32)
    T_cw = np.linalg.inv(T_wc)
    for idx, track in enumerate(tracks):
        track = track.copy()
        tmp = np.zeros((len(track), 79)) - 1
        box = np.clip(track[-1, -4:] / np.array([640, 480, 640, 480]), a_min=-1, a_max=2)
        track[:, 2:6] = box
        tmp[:, :9] = track[:, :9]
        tmp[:, 9:12] = (get_homogeneous(track[:, 9:12]) @ T_cw.T)[:, :3]
        ang = track[:, 12] - cam_azi
        tmp[:, 12] = np.sin(ang); tmp[:, 13] = np.cos(ang); tmp[:, 14] = track[:, 13]; tmp[:, 15:79] = track[:, 14:78]
        tmp = tmp.astype(np.float32)
        if len(track) > 100:
            want[idx] = tmp[-100:]
        else:
            want[idx, :len(track)] = tmp
    assert np.array_equal(got, want)
