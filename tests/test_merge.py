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


def test_merge_leaves_its_input_tracks_untouched(golden):
    """the reference picks rows with a boolean mask (a copy, run_merge.py:36): the merged class never reaches the
    caller's tracks"""
    from odam_amd import merge
    z, tracks, _ = _load(golden)
    tracks[0][:, 1] = 4; tracks[1][:, 1] = 5; tracks[1][:5, 1] = 4      # a sofa/chair pair with a mixed label history
    before = [t.copy() for t in tracks]
    merge.merge_process({"tracks": tracks, "bboxes_qc": list(z["bboxes_qc"])}, [int(x) for x in z["img_names"]])
    for a, b in zip(tracks, before):
        assert np.array_equal(a, b)


def test_pair_iou_matches_the_reference(golden):
    """512 box pairs through the reference's box3d_iou (golden): overlapping, contained, disjoint, no z overlap, the
    clip rectangle wound the other way (reference: 0) and all pairs of fitted boxes"""
    from odam_amd import merge
    z = golden("box_iou.npz")
    i3, i2 = merge.box3d_iou_pairs(z["A"], z["B"])
    assert np.abs(i3 - z["iou3d"]).max() <= 1e-12 and np.abs(i2 - z["iou_bev"]).max() <= 1e-12
    assert np.array_equal(i3 == 0, z["iou3d"] == 0) and (z["iou3d"] == 0).sum() > 50 and (z["iou3d"] > 0.3).sum() > 50
    for k in (0, 100, 400):
        assert merge.box3d_iou(z["A"][k], z["B"][k]) == (i3[k], i2[k])


def test_cost_matrix_at_config5_size():
    """500 objects (125 k pairs): symmetric, zero diagonal, in [0, 1], well under a second"""
    import time
    from odam_amd import merge
    from odam_amd.multi_view import get_3d_box
    rs = np.random.RandomState(0)

    def rotz(t):
        c, s = np.cos(t), np.sin(t)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    boxes = [get_3d_box(rs.uniform(.3, 2, 3), rotz(rs.uniform(-3, 3)), rs.uniform(-4, 4, 3)) for _ in range(500)]
    tracks = [np.tile(np.r_[0., rs.randint(0, 8), np.zeros(80)], (3, 1)) for _ in range(500)]
    t = time.time()
    c = merge.cost_matrix(tracks, boxes)
    dt = time.time() - t
    assert dt < 2.0, dt
    assert np.array_equal(c, c.T) and (np.diag(c) == 0).all() and c.min() >= 0 and c.max() <= 1 and (c < 1).sum() > 100


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
