"""Multi-view super-quadric fitting of object tracks -- host driver over the gfx950 kernel.

Drop-in for the reference's `optim_process` (likojack/ODAM src/scripts/run_multi_view.py:22-76):
same arguments, same output dict {"tracks", "bboxes_qc", "bboxes_dl", "quadrics"}.  What changes
is the schedule: the reference fits one object at a time on the CPU (autograd, 200 steps); here
the host prelude is vectorised per object and ALL objects with enough views go to the GPU in one
`odam_sq_fit_batch` launch (one workgroup per object).

Reference pieces restated here (host, float64 numpy, same library calls where a library defines
the result):
  src/utils/tracking_gt_utils.py:145-211  load_pred_object   -> _object_constraints
  src/utils/tracking_gt_utils.py:59-66    averaging_T_wos    -> scipy Rotation.mean (same call)
  src/super_quadric/quadric_helper.py:69-109  bbox_to_lines  -> _edge_lines
  src/utils/box_utils.py:286-308          get_3d_box
  src/utils/box_utils.py:319-410          compute_oriented_bbox (qhull hull + rotating calipers)
"""
import math

import numpy as np
from scipy.spatial import ConvexHull
from scipy.spatial.transform import Rotation

from . import sq as _sq

EDGE_THRESHOLD = 20  # tracking_gt_utils.py:199


def rotz(t):  # box_utils.py:311-316
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def get_3d_box(box_size, rot_mat, center):
    """8 corners of an oriented box (box_utils.py:286-308)."""
    l, w, h = box_size
    x = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    z = [h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2]
    c = np.dot(rot_mat, np.vstack([x, y, z]))
    c[0, :] += center[0]
    c[1, :] += center[1]
    c[2, :] += center[2]
    return np.transpose(c)


def compute_oriented_bbox(pts):
    """Minimum-area upright box of a point set (box_utils.py:319-410).

    Quirks kept: the hull polygon is NOT closed (the edge from the last vertex back to the first is
    never tested), angles are folded into [0, pi/2) and de-duplicated, the first smallest area wins.
    """
    z_min = np.min(pts[:, 2])
    z_max = np.max(pts[:, 2])
    pts_xy = pts[:, :2]
    hull = ConvexHull(pts_xy)
    contour = pts_xy[hull.vertices].astype(pts_xy.dtype, copy=True)
    x_mean, y_mean = np.mean(contour, axis=0)
    contour[:, 0] -= x_mean
    contour[:, 1] -= y_mean
    edges = (contour[1:] - contour[:-1]).astype(np.float64)
    angles = np.array([abs(math.atan2(e[1], e[0]) % (math.pi / 2)) for e in edges])
    angles = np.unique(angles)
    # all candidate rotations at once (the reference loops; np.matmul runs the same 2x2 @ 2xV product per angle)
    ca = np.array([[math.cos(a), math.cos(a - (math.pi / 2)), math.cos(a + (math.pi / 2))] for a in angles]).reshape(-1, 3)
    Rs = np.stack([ca[:, [0, 1]], ca[:, [2, 0]]], axis=1)                 # [A, 2, 2]
    rot = np.matmul(Rs, np.transpose(contour))                            # [A, 2, V]
    mins, maxs = np.nanmin(rot, axis=2), np.nanmax(rot, axis=2)           # [A, 2]
    areas = (maxs[:, 0] - mins[:, 0]) * (maxs[:, 1] - mins[:, 1])
    best = (0, 10000000000, 0, 0, 0, 0)
    ok = areas < best[1]                                                  # NaN areas never win
    if ok.any():
        i = int(np.flatnonzero(areas == areas[ok].min())[0])              # the first smallest area wins
        best = (angles[i], areas[i], mins[i, 0], maxs[i, 0], mins[i, 1], maxs[i, 1])
    a, _, min_x, max_x, min_y, max_y = best
    R = np.array([[math.cos(a), math.cos(a - (math.pi / 2))], [math.cos(a + (math.pi / 2)), math.cos(a)]])
    c2 = np.zeros((4, 2))
    c2[0] = np.dot([max_x, max_y], R)
    c2[1] = np.dot([max_x, min_y], R)
    c2[2] = np.dot([min_x, min_y], R)
    c2[3] = np.dot([min_x, max_y], R)
    c2[:, 0] += x_mean
    c2[:, 1] += y_mean
    upper = np.concatenate([c2, np.array([[z_max] * 4]).T], axis=1)
    lower = np.concatenate([c2, np.array([[z_min] * 4]).T], axis=1)
    return np.concatenate([upper, lower], axis=0)


def compute_oriented_bboxes(points):
    """compute_oriented_bbox for a batch: points [n, N, 3] float32 -> list of n [8, 3] float64 boxes.

    The library's host routine (include/odam_sq.h, odam_sq_oriented_bbox; csrc/hull2d.h) follows qhull's 2-D construction so that
    the hull's vertex ORDER -- which decides the edge the reference's open polygon leaves out -- is qhull's; an object that sits
    inside qhull's round-off tolerance band (status 1: exact coincidences, cube-like surfaces) is recomputed here with
    scipy's qhull, the library the reference itself calls.  Returns (boxes, n_recomputed)."""
    import ctypes
    from . import _lib
    pts = np.ascontiguousarray(points, np.float32)
    n = len(pts)
    if n == 0:
        return [], 0
    corners = np.empty((n, 8, 3), np.float64)
    status = np.ones(n, np.int32)
    _lib.check(_lib.lib().odam_sq_oriented_bbox(pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n), ctypes.c_int(pts.shape[1]),
                                                corners.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.c_void_p)),
               "odam_sq_oriented_bbox")
    redo = np.flatnonzero(status)
    for i in redo:
        corners[i] = compute_oriented_bbox(pts[i])
    return [corners[i] for i in range(n)], len(redo)


def _edge_lines(bbox, img_h, img_w, thr=EDGE_THRESHOLD):
    """bbox [[x_min,y_min],[x_max,y_max]] -> {name: [a, b, -value]} for edges farther than `thr`
    pixels from the image border (quadric_helper.py:69-109)."""
    (x_min, y_min), (x_max, y_max) = bbox
    out = {}
    for name, v, lim in (("x_min", x_min, img_w), ("y_min", y_min, img_h),
                         ("x_max", x_max, img_w), ("y_max", y_max, img_h)):
        if v > thr and v < lim - thr:
            out[name] = np.array([1, 0, -v]) if name[0] == "x" else np.array([0, 1, -v])
    return out


def _object_constraints(track, frame_to_img, img_h, img_w, thr=EDGE_THRESHOLD):
    """Per observed frame of one track, in image order, vectorised over the observations.

    tracking_gt_utils.py:145-211 walks every image, tests membership and builds per-frame dicts; here the
    membership is one dictionary lookup per observation (first row of a frame id wins, as
    np.where(...)[0][0] does) and the bbox-edge constraints go straight into arrays:
      tgt[n,4], mask[n,4] in the order x_min, x_max, y_min, y_max (sq_libs.py:438): mask = edge farther
      than `thr` px from the border (quadric_helper.py:87-107); tgt = float32(-float32(-pixel)), i.e. what
      `-gt` is after sq_libs.py:448 stores the line's last entry in a float32 tensor.
    Returns class, image ids [n], tgt, mask, rotation matrices [n,3,3], t_wo [3], dims [n,3].
    """
    frames = track[:, 0].astype(np.int32)
    if isinstance(frame_to_img, _UniqueFrames):
        # every frame id names ONE image (the usual sequence): first row of each id by np.unique, image index by binary search,
        # image order by one argsort -- the same (img_id, row) list as the walk below, without a Python loop per observation
        uniq, first = np.unique(frames, return_index=True)
        pos = np.searchsorted(frame_to_img.ids, uniq)
        pos[pos >= len(frame_to_img.ids)] = 0
        hit = frame_to_img.ids[pos] == uniq
        img_ids = frame_to_img.order[pos[hit]]
        rows = first[hit].astype(np.int64)
        o = np.argsort(img_ids, kind="stable")
        img_ids, rows = img_ids[o], rows[o]
    else:
        first_row = {}
        for r, fid in enumerate(frames.tolist()):
            first_row.setdefault(fid, r)
        obs = sorted((img_id, r) for fid, r in first_row.items() for img_id in frame_to_img.get(fid, ()))
        img_ids = np.array([o[0] for o in obs], np.int64)
        rows = np.array([o[1] for o in obs], np.int64)
    obj_class = int(np.median(track[:, 1]))
    t_wo = np.mean(track[:, 9:12], axis=0)
    sub = track[rows]
    c, s_ = np.cos(sub[:, 12]), np.sin(sub[:, 12])          # box_utils.rotz per observation
    R = np.zeros((len(rows), 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s_, s_, c, 1.0
    vals = sub[:, [2, 4, 3, 5]]                              # x_min, x_max, y_min, y_max
    lims = np.array([img_w, img_w, img_h, img_h], np.float64)
    mask = ((vals > thr) & (vals < lims - thr)).astype(np.float32)
    tgt = np.where(mask > 0, (-((-vals).astype(np.float32))), np.float32(0)).astype(np.float32)
    return obj_class, img_ids, tgt, mask, R, t_wo, sub[:, 6:9]


class _UniqueFrames:
    """frame id -> image index for a sequence whose frame ids are all different: ids sorted, order[i] = image index of ids[i]"""

    def __init__(self, img_names):
        names = np.asarray([int(f) for f in img_names], np.int64)
        self.order = np.argsort(names, kind="stable").astype(np.int64)
        self.ids = names[self.order]

    @staticmethod
    def applies(img_names):
        names = np.asarray([int(f) for f in img_names], np.int64)
        return len(names) > 0 and len(np.unique(names)) == len(names) and np.abs(names).max() < 2 ** 31


def averaging_T_wos_batch(R_list, t_list):
    """averaging_T_wos for many objects with ONE conversion of all rotation matrices, one batched symmetric eigen-decomposition and
    one conversion back -- the same operations scipy's Rotation.mean applies per object (quaternions q of the observations,
    K = q^T q by np.dot, eigenvector of the largest eigenvalue from np.linalg.eigh, taken as the mean quaternion WITHOUT
    renormalising), so every matrix equals the per-object call bit for bit (tests/test_multi_view_host.py checks 400 objects)."""
    n = len(R_list)
    out = np.tile(np.eye(4), (n, 1, 1))
    if n == 0:
        return out
    lens = [len(R) for R in R_list]
    q_all = Rotation.from_matrix(np.concatenate(R_list)).as_quat()
    K = np.empty((n, 4, 4))
    o = 0
    for i, k in enumerate(lens):
        q = q_all[o:o + k]
        K[i] = np.dot(q.T, q)
        o += k
    _, v = np.linalg.eigh(K)
    out[:, :3, :3] = Rotation(np.ascontiguousarray(v[:, :, -1]), normalize=False).as_matrix()
    for i, (t, k) in enumerate(zip(t_list, lens)):
        out[i, :3, 3] = np.mean(np.repeat(t[None, :], k, axis=0), axis=0)
    return out


def averaging_T_wos(R_wos, t_wo):
    """tracking_gt_utils.py:59-66 for T_wos that all carry the same translation (load_pred_object gives every
    per-frame T_wo the track's mean t_wo, :148/:186-189): mean rotation by scipy, mean of n equal vectors."""
    out = np.eye(4)
    out[:3, :3] = Rotation.from_matrix(R_wos).mean().as_matrix()
    out[:3, 3] = np.mean(np.repeat(t_wo[None, :], len(R_wos), axis=0), axis=0)
    return out


class SuperQuadric:
    """Result object with the attribute surface of the reference's SuperQuadric
    (sq_libs.py:531-595): translate[3], angle[], scales[3], shapes[2] as float32 numpy, obj_class,
    and compute_ellipsoid_points(use_numpy) -> ([1000,3] float32, None)."""

    def __init__(self, params, obj_class, points=None, fitter=None):
        p = np.asarray(params, np.float32)
        self.translate = p[0:3].copy()
        self.angle = np.float32(p[3])
        self.scales = p[4:7].copy()
        self.shapes = p[7:9].copy()
        self.obj_class = obj_class
        self._points = points
        self._fitter = fitter

    @property
    def params(self):
        return np.concatenate([self.translate, [self.angle], self.scales, self.shapes]).astype(np.float32)

    # run_processor.py:85-92 pickles the result dict; eval_scan2cad.py:191-215 and result_viewer.py:25-43 load it and
    # call compute_ellipsoid_points.  The device context does not travel: a loaded object without cached points
    # computes them with the default fitter.
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_fitter"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    def compute_ellipsoid_points(self, use_numpy=True):
        if self._points is None:
            fitter = self._fitter or default_fitter()
            self._points = fitter.points(self.params[None])[0].cpu().numpy()
        return self._points, None


_DEFAULT_FITTER = {}


def default_fitter(device="cuda:0"):
    if device not in _DEFAULT_FITTER:
        _DEFAULT_FITTER[device] = _sq.SqFitter(device, 200)
    return _DEFAULT_FITTER[device]


def optim_process(tracks, img_names, T_wcs, P_cws, img_h, img_w, K, representation, prior, n_iters, n_views,
                  fitter=None, return_params=False):
    """run_multi_view.py:22-76 with the per-object fits batched on the GPU."""
    fitter = fitter or default_fitter()
    n_objs = len(tracks)
    if _UniqueFrames.applies(img_names):
        frame_to_img = _UniqueFrames(img_names)
    else:
        frame_to_img = {}
        for i, fid in enumerate(img_names):
            frame_to_img.setdefault(int(fid), []).append(i)  # every image whose id is in the track
    P_all = np.asarray(P_cws)

    inits, classes, bboxes_dl, fit_ids = [], [], [], []
    fit_P, fit_tgt, fit_mask, fit_counts = [], [], [], []
    cons = [_object_constraints(np.asarray(tracks[obj_id]), frame_to_img, img_h, img_w) for obj_id in range(n_objs)]
    # mean pose of every object (tracking_gt_utils.py:59-66) and its yaw: the scipy conversions once for all objects
    if n_objs and all(len(c[4]) for c in cons):
        T_wos = averaging_T_wos_batch([c[4] for c in cons], [c[5] for c in cons])
        yaws = Rotation.from_matrix(T_wos[:, :3, :3]).as_euler("zxy")[:, 0]
    else:       # an object without an observed frame: the per-object calls (and their errors) as they were
        T_wos = yaws = None
    for obj_id in range(n_objs):
        obj_class, img_ids, tgt, mask, R_wos, t_wo, dims = cons[obj_id]
        T_wo = T_wos[obj_id] if T_wos is not None else averaging_T_wos(R_wos, t_wo)
        scales = np.mean(np.asarray(dims), axis=0)
        bboxes_dl.append(get_3d_box(scales, T_wo[:3, :3], T_wo[:3, 3]))
        yaw = yaws[obj_id] if yaws is not None else Rotation.from_matrix(T_wo[:3, :3]).as_euler("zxy")[0]
        if prior and obj_class not in _sq.CLASS_MAPPER:
            raise KeyError(obj_class)  # sq_libs.py:464 (CLASS_MAPPER covers classes 0..7 only)
        inits.append(_sq.init_params(T_wo[:3, 3], yaw, scales, representation))
        classes.append(obj_class)
        valid = mask.any(axis=1)       # frames with at least one constrained edge (run_multi_view.py:51-54)
        if int(valid.sum()) >= n_views:
            fit_ids.append(obj_id)
            fit_P.append(P_all[img_ids[valid]].astype(np.float32).reshape(-1, 12))
            fit_tgt.append(tgt[valid])
            fit_mask.append(mask[valid])
            fit_counts.append(int(valid.sum()))

    params = {i: inits[i] for i in range(n_objs)}
    points = {}
    if fit_ids:
        out = fitter.fit(np.stack([inits[i] for i in fit_ids]), [classes[i] for i in fit_ids], fit_counts,
                         np.concatenate(fit_P), np.concatenate(fit_tgt), np.concatenate(fit_mask),
                         n_iters=n_iters, representation=representation, prior=bool(prior), want_points=True)
        fp = out["params"].cpu().numpy()
        fpts = out["points"].cpu().numpy()
        for j, i in enumerate(fit_ids):
            params[i] = fp[j]
            points[i] = fpts[j]

    quadrics, bboxes_qc = [], []
    boxes = {}
    if fit_ids:       # oriented boxes of all fitted surfaces in one native call (run_multi_view.py:66-67)
        bl, _ = compute_oriented_bboxes(np.stack([points[i] for i in fit_ids]))
        boxes = dict(zip(fit_ids, bl))
    for obj_id in range(n_objs):
        if obj_id in points:  # run_multi_view.py:64-69
            quadrics.append(SuperQuadric(params[obj_id], classes[obj_id], points[obj_id], fitter))
            bboxes_qc.append(boxes[obj_id])
        else:                 # fewer than n_views constrained frames: keep the detector box (:59-62)
            quadrics.append(SuperQuadric(params[obj_id], classes[obj_id], None, fitter))
            bboxes_qc.append(bboxes_dl[obj_id])
    out_dict = {"tracks": tracks, "bboxes_qc": bboxes_qc, "bboxes_dl": bboxes_dl, "quadrics": quadrics}
    if return_params:
        out_dict["params"] = np.stack([params[i] for i in range(n_objs)]) if n_objs else np.zeros((0, 9), np.float32)
        out_dict["fitted"] = np.array([i in points for i in range(n_objs)], bool)
    return out_dict
