"""odam_amd.merge against the reference's merge_process / box3d_iou (golden from tests/golden/make_golden.py merge)."""
import numpy as np


def _load(golden):
    z = golden("sq_merge.npz")
    tracks = [z[f"track{i}"].copy() for i in range(int(z["n_tracks"]))]
    merged = [z[f"merged{i}"] for i in range(int(z["n_merged"]))]
    return z, tracks, merged


def test_merge_process_matches_reference(golden):
    from odam_amd import merge
    z, tracks, ref = _load(golden)
    out = merge.merge_process({"tracks": tracks, "bboxes_qc": list(z["bboxes_qc"])}, [int(x) for x in z["img_names"]])
    assert len(tracks) == 9 and len(ref) == 5            # fragments of one object were merged back
    assert len(out) == len(ref)
    for a, b in zip(out, ref):
        assert a.shape == b.shape
        assert np.array_equal(a, b)                        # rows are copied observations: exact


def test_cost_matrix_shortcut_equals_full_pair_loop(golden):
    """the bounds pre-test only skips pairs whose clipped polygon would be empty"""
    from odam_amd import merge
    z, tracks, _ = _load(golden)
    boxes = list(z["bboxes_qc"])
    fast = merge.cost_matrix(tracks, boxes)
    n = len(tracks)
    full = np.zeros((n, n))
    cls = [int(np.median(t[:, 1])) for t in tracks]
    for i in range(n):
        for j in range(i + 1, n):
            full[i, j] = 1 - merge.box3d_iou(boxes[i], boxes[j])[0] if merge._mergeable(cls[i], cls[j]) else 1
    full += full.T
    assert np.array_equal(fast, full)
    assert (fast < 0.95).sum() >= 4                        # the scene really contains mergeable pairs


def test_box3d_iou_basic():
    from odam_amd import merge
    from odam_amd.multi_view import get_3d_box
    a = get_3d_box([1.0, 2.0, 1.0], np.eye(3), [0, 0, 0.5])
    assert abs(merge.box3d_iou(a, a)[0] - 1.0) < 1e-12
    b = get_3d_box([1.0, 2.0, 1.0], np.eye(3), [0.5, 0, 0.5])
    iou3, iou2 = merge.box3d_iou(a, b)
    assert abs(iou2 - 1.0 / 3.0) < 1e-12 and abs(iou3 - 1.0 / 3.0) < 1e-12
    c = get_3d_box([1.0, 2.0, 1.0], np.eye(3), [5.0, 0, 0.5])
    assert merge.box3d_iou(a, c)[0] == 0.0
