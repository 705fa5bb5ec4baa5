"""Track merging between the two super-quadric passes -- SURVEY.md section 8(f) rank 4.

`merge_process` keeps the behaviour of the reference's (likojack/ODAM src/scripts/run_merge.py:79-130): cost
= 1 - 3D IoU of the fitted boxes for mergeable class pairs, sklearn AgglomerativeClustering on the precomputed
cost (average linkage, threshold 0.95), per image the observation of the longest member track,
scipy.stats.mode for the merged class.

The pair cost is not the reference's per-pair Python polygon clipper + qhull (box_utils.py:24-120) but a closed
form for two convex quadrilaterals evaluated for all mergeable pairs at once: the intersection's boundary is
made of the pieces of either polygon's edges that lie inside the other polygon; each edge is cut against the
other polygon's four half-planes parametrically and the area is the boundary integral of (x dy - y dx) / 2.
Config-5 sizes (500 objects, 125 k pairs) take milliseconds.  Values agree with the reference's box3d_iou to
rounding (tests/golden/box_iou.npz), clusters are identical (tests/golden/sq_merge.npz).

`AgglomerativeClustering(affinity=...)` of the reference is spelled `metric=` in scikit-learn >= 1.2.
"""
import numpy as np
import scipy.stats
from sklearn.cluster import AgglomerativeClustering


def _boundary_inside(P, Q, closed):
    """Green's-theorem contribution of the parts of P's edges that lie inside the convex quadrilateral Q.

    P, Q: [n, 4, 2] counter-clockwise vertex lists.  Each edge p -> q of P is cut against the four half-planes of Q
    parametrically (enter = largest t at which the edge crosses into a half-plane, leave = smallest t at which it
    crosses out); the surviving piece [enter, leave] contributes cross(p(enter), p(leave)) / 2.  `closed` says whether
    an edge lying exactly on Q's boundary counts as inside (it must count for one of the two polygons only)."""
    p = P                                            # [n, 4, 2]
    d = np.roll(P, -1, axis=1) - P                   # edge vectors
    a = Q[:, None, :, :]                             # [n, 1, 4, 2] half-plane anchors
    e = (np.roll(Q, -1, axis=1) - Q)[:, None, :, :]  # half-plane directions; inside = left of a -> a + e
    pp = p[:, :, None, :]; dd = d[:, :, None, :]
    dist = e[..., 0] * (pp[..., 1] - a[..., 1]) - e[..., 1] * (pp[..., 0] - a[..., 0])   # [n, 4 edges, 4 planes]
    rate = e[..., 0] * dd[..., 1] - e[..., 1] * dd[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -dist / rate
    par = rate == 0
    out_par = par & ((dist < 0) if closed else (dist <= 0))       # parallel and on the wrong side: nothing survives
    enter = np.where((rate > 0) & ~par, t, -np.inf).max(axis=2)
    leave = np.where((rate < 0) & ~par, t, np.inf).min(axis=2)
    t0 = np.clip(enter, 0.0, 1.0); t1 = np.clip(leave, 0.0, 1.0)
    ok = (t1 > t0) & ~out_par.any(axis=2)
    s = p + t0[..., None] * d; q = p + t1[..., None] * d
    return np.where(ok, s[..., 0] * q[..., 1] - s[..., 1] * q[..., 0], 0.0).sum(axis=1) * 0.5


def _signed_area(R):
    x, y = R[..., 0], R[..., 1]
    return 0.5 * (x * np.roll(y, -1, axis=-1) - y * np.roll(x, -1, axis=-1)).sum(axis=-1)


def rect_intersection_area(R1, R2):
    """Area of the intersection of convex quadrilaterals R1[i] and R2[i] ([n, 4, 2], any winding), all pairs at
    once: the boundary of the intersection is {edges of R1 inside R2} + {edges of R2 inside R1}, and the area is
    the line integral of (x dy - y dx) / 2 along it."""
    s1 = _signed_area(R1); s2 = _signed_area(R2)
    A = np.where((s1 < 0)[:, None, None], R1[:, ::-1], R1)
    B = np.where((s2 < 0)[:, None, None], R2[:, ::-1], R2)
    return np.maximum(_boundary_inside(A, B, True) + _boundary_inside(B, A, False), 0.0)


def box3d_iou_pairs(C1, C2):
    """3D IoU and bird's-eye IoU of box pairs (C1[i], C2[i]), [n, 8, 3] corner arrays in the layout of
    `compute_oriented_bbox` / `get_3d_box` (top face 0-3, bottom face 4-7) -- the quantity of the reference's
    box3d_iou (src/utils/box_utils.py:98-120), evaluated for all pairs at once.  The reference clips rectangle 1
    (corners 3,2,1,0) by rectangle 2 with a clipper that assumes rectangle 2 is counter-clockwise in that order
    and returns nothing otherwise; that case (a box whose top face winds the other way) keeps its IoU of 0."""
    C1 = np.asarray(C1, np.float64); C2 = np.asarray(C2, np.float64)
    R1 = C1[:, 3::-1, :2]; R2 = C2[:, 3::-1, :2]
    a1 = np.abs(_signed_area(R1)); a2 = np.abs(_signed_area(R2))
    inter = np.where(_signed_area(R2) > 0, rect_intersection_area(R1, R2), 0.0)
    iou_bev = inter / (a1 + a2 - inter)
    dz = np.maximum(0.0, np.minimum(C1[:, 0, 2], C2[:, 0, 2]) - np.maximum(C1[:, 4, 2], C2[:, 4, 2]))

    def vol(c):
        return (np.linalg.norm(c[:, 0] - c[:, 1], axis=1) * np.linalg.norm(c[:, 1] - c[:, 2], axis=1)
                * np.linalg.norm(c[:, 0] - c[:, 4], axis=1))
    iv = inter * dz
    return iv / (vol(C1) + vol(C2) - iv), iou_bev


def box3d_iou(c1, c2):
    """one pair: (iou_3d, iou_bev)"""
    a, b = box3d_iou_pairs(np.asarray(c1)[None], np.asarray(c2)[None])
    return float(a[0]), float(b[0])


def cost_matrix(tracks, bboxes_qc):
    """run_merge.py:92-118: 1 - 3D IoU for mergeable class pairs, 1 otherwise; symmetric, zero diagonal.
    All mergeable pairs (i < j) go through one vectorised IoU evaluation."""
    n = len(tracks)
    cls = np.array([int(np.median(t[:, 1])) for t in tracks])
    boxes = np.asarray([np.asarray(b, np.float64) for b in bboxes_qc]).reshape(n, 8, 3)
    sofa_chair = (cls == 4) | (cls == 5)             # run_merge.py:105-108
    ok = (cls[:, None] == cls[None, :]) | (sofa_chair[:, None] & sofa_chair[None, :])
    i, j = np.nonzero(np.triu(ok, 1))
    cost = np.triu(np.ones((n, n)), 1)
    if len(i):
        cost[i, j] = 1 - box3d_iou_pairs(boxes[i], boxes[j])[0]
    return cost + cost.T


def _merge_cluster(tracks, mask, img_names):
    """run_merge.py:24-58: per image keep the observation of the longest member track; class = mode"""
    members = [i for i in range(len(tracks)) if mask[i]]
    dom = np.concatenate([tracks[i][:, 1] for i in members], axis=0)
    dom = int(scipy.stats.mode(dom).mode)
    by_frame = {}
    for i in members:
        for row in tracks[i]:
            by_frame.setdefault(row[0], []).append((i, row))
    out = []
    for name in img_names:
        cands = by_frame.get(name)
        if not cands:
            continue
        # np.argmax over track lengths in member order: first longest wins
        best = max(range(len(cands)), key=lambda k: (len(tracks[cands[k][0]]), -k))
        row = cands[best][1].copy()   # the reference selects rows with a boolean mask (a copy, run_merge.py:36):
        row[1] = dom                  # the merged class goes into the copy, the input tracks stay as they were
        out.append(row)
    return np.asarray(out)


def merge_process(data, img_names):
    """run_merge.py:79-130: data = optim_process output dict; returns the list of merged [n,82] tracks."""
    tracks = data["tracks"]
    n = len(tracks)
    if n == 1:
        merged = tracks
    else:
        cost = cost_matrix(tracks, data["bboxes_qc"])
        labels = AgglomerativeClustering(n_clusters=None, distance_threshold=0.95, metric="precomputed",
                                         linkage="average").fit(cost).labels_
        merged = [_merge_cluster(tracks, labels == c, img_names) for c in np.unique(labels)]
    return [t for t in merged if len(t) > 0]
