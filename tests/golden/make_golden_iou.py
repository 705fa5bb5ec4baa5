"""Golden for the merge cost: the reference's box3d_iou (src/utils/box_utils.py:98-120, imported here) on pairs of
oriented boxes -- random overlapping pairs, contained, identical, disjoint, touching-at-a-corner, one box with the
opposite winding, and all pairs of the fitted boxes of sq_merge.npz.  Stores inputs + (iou_3d, iou_bev).
Run: python tests/golden/make_golden_iou.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402


def main():
    import refenv
    refenv.setup()
    import src.utils.box_utils as bu
    rs = np.random.RandomState(11)
    A, B = [], []

    def box(c, dims, yaw):
        return bu.get_3d_box(dims, bu.rotz(yaw), c)
    for _ in range(300):
        c = rs.uniform(-1, 1, 3); d = rs.uniform(0.3, 2.0, 3); y = rs.uniform(-np.pi, np.pi)
        c2 = c + rs.normal(0, 0.6, 3); d2 = d * rs.uniform(0.5, 1.6, 3); y2 = y + rs.normal(0, 0.7)
        A.append(box(c, d, y)); B.append(box(c2, d2, y2))
    for _ in range(30):                                   # contained / identical / far apart / axis-aligned offsets
        c = rs.uniform(-1, 1, 3); d = rs.uniform(0.5, 2.0, 3); y = rs.uniform(-np.pi, np.pi)
        A.append(box(c, d, y)); B.append(box(c, d * 0.5, y + 0.3))
        A.append(box(c, d, y)); B.append(box(c + 9.0, d, y))
        A.append(box(c, d, 0.0)); B.append(box(c + np.r_[d[0] * 0.25, 0, 0], d, 0.02))   # (exactly collinear edges: the reference's strict-inequality clipper is rounding noise there)
        A.append(box(c, d, y)); B.append(box(c + np.r_[0.01, 0, d[2] * 2], d, y + 0.01))   # footprints overlap, no z overlap
    for _ in range(10):                                   # clip rectangle wound the other way: the reference gives 0
        c = rs.uniform(-1, 1, 3); d = rs.uniform(0.5, 2.0, 3); y = rs.uniform(-np.pi, np.pi)
        b2 = box(c + 0.1, d, y + 0.2)
        A.append(box(c, d, y)); B.append(b2[[3, 2, 1, 0, 7, 6, 5, 4]])
        A.append(b2[[3, 2, 1, 0, 7, 6, 5, 4]]); B.append(box(c, d, y))
    z = np.load(os.path.join(HERE, "sq_merge.npz"))
    qc = z["bboxes_qc"]
    for i in range(len(qc)):
        for j in range(len(qc)):
            if i != j:
                A.append(qc[i]); B.append(qc[j])
    A = np.asarray(A); B = np.asarray(B)
    out, keep = [], []
    for k, (a, b) in enumerate(zip(A, B)):   # exactly coincident edges divide by zero in the reference's clipper: not a case
        try:
            with np.errstate(all="raise"):
                out.append(bu.box3d_iou(a, b)); keep.append(k)
        except (ValueError, FloatingPointError):
            pass
    out = np.asarray(out); A = A[keep]; B = B[keep]
    np.savez_compressed(os.path.join(HERE, "box_iou.npz"), A=A, B=B, iou3d=out[:, 0], iou_bev=out[:, 1])
    print("box iou golden:", len(A), "pairs; nonzero", int((out[:, 0] > 0).sum()))


if __name__ == "__main__":
    main()
