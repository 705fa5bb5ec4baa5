"""Host side of the super-quadric multi-view fit: thin Python over the C ABI (include/odam_sq.h).

Mirrors the pieces of the reference (likojack/ODAM) that sit around SuperQuadricOptimizer.run:
  src/super_quadric/sq_libs.py:353-393  (init: scales -> sqrt(dims/2), shapes = -0 / -10000, prior table)
  src/super_quadric/sq_libs.py:438-451  (gt / mask arrays per direction)
  src/super_quadric/sq_libs.py:13-22    (CLASS_MAPPER order of the prior rows)
"""
import ctypes
import os
import pickle
import threading

import numpy as np
import torch

from . import _lib

N_POINTS = 1000
WG_VIEWS = 1024          # rows one workgroup reduces (ODAM_SQ_MAX_VIEWS)
MAX_VIEWS = 16 * WG_VIEWS  # per object: split over up to 16 workgroups
NAMES = ("x_min", "x_max", "y_min", "y_max")  # sq_libs.py:438
REPRESENTATIONS = {"super_quadric": 0, "cube": 1, "quadric": 2}
# sq_libs.py:13-22
CLASS_MAPPER = {0: "03211117", 1: "04379243", 2: "02808440", 3: "02747177",
                4: "04256520", 5: "03001627", 6: "02933112", 7: "02871439"}

_PRIOR_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scale_prior.npz")


def load_scale_prior(path=None):
    """[8, 9] float32 inverse covariances in CLASS_MAPPER order.

    `path` may be the reference's pickle (src/super_quadric/scale_prior, a dict synset -> 3x3
    float64) or None for the copy of those 72 numbers shipped in odam_amd/data/scale_prior.npz.
    """
    if path is None:
        z = np.load(_PRIOR_PATH)
        return z["icov"].astype(np.float32).reshape(8, 9)
    with open(path, "rb") as f:
        d = pickle.load(f)
    return np.stack([np.asarray(d[CLASS_MAPPER[k]], np.float64).astype(np.float32).reshape(9)
                     for k in range(8)])


def lines_to_targets(bbox_lines):
    """list (per valid frame) of {name: [a, b, -pixel]} -> tgt[F,4], mask[F,4] float32 (sq_libs.py:438-451)."""
    F = len(bbox_lines)
    tgt = np.zeros((F, 4), np.float32)
    mask = np.zeros((F, 4), np.float32)
    for f, d in enumerate(bbox_lines):
        for k, name in enumerate(NAMES):
            if name in d:
                mask[f, k] = 1.0
                # gt = float32(line[-1]) = -pixel; the residual compares with -gt
                tgt[f, k] = -np.float32(d[name][-1])
    return tgt, mask


def init_params(translate, angle, dims, representation="super_quadric"):
    """SuperQuadricOptimizer.__init__ (sq_libs.py:353-371): 9 float32 parameters."""
    scales = np.sqrt(np.asarray(dims, np.float64) / 2)
    shapes = np.array([-10000.0, -10000.0]) if representation == "cube" else np.array([-0.0, -0.0])
    return np.concatenate([np.asarray(translate, np.float64).reshape(3), [float(angle)], scales,
                           shapes]).astype(np.float32)


class SqFitter:
    """Owns the device context (constant sampler draws + Adam tables)."""

    def __init__(self, device="cuda:0", max_iters=200):
        self.device = torch.device(device)
        self.max_iters = int(max_iters)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            h = ctypes.c_void_p()
            _lib.check(L.odam_sq_create(ctypes.c_int(self.max_iters), ctypes.byref(h)), "odam_sq_create")
        self._h = h
        self._prior = None
        # The native handle serves ONE launch at a time (include/odam_sq.h): its view-split exchange buffer is per
        # handle.  The lock serialises host threads; launches from different streams are additionally ordered by an
        # event so that a second fit cannot start writing exchange slots while the first still polls them.
        self._lock = threading.Lock()
        self._last = None          # (stream id, event) of the most recent launch
        with torch.cuda.device(self.device):
            self.n_cu = int(torch.cuda.get_device_properties(self.device).multi_processor_count)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().odam_sq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prior_dev(self):
        if self._prior is None:
            self._prior = torch.from_numpy(load_scale_prior()).to(self.device).contiguous()
        return self._prior

    def fit(self, params0, class_ids, view_counts, P, tgt, mask, n_iters=200, representation="super_quadric",
            prior=True, want_points=True, want_loss=False, want_traj=False):
        """Batched SuperQuadricOptimizer.run.

        params0 [n,9] f32; class_ids [n] int; view_counts [n] int; P [sumF,3,4]/[sumF,12] f32;
        tgt, mask [sumF,4] f32 (numpy or torch, host or device).  Returns dict of device tensors.

        Objects with more than 1024 views need the library's view split (k workgroups per object, k x padded object
        count <= number of CUs): they are fitted in groups small enough for that; fits are independent, so the
        grouping does not change any result.
        """
        vc = np.asarray(view_counts, np.int64)
        if len(vc) and vc.max() > WG_VIEWS:
            return self._fit_grouped(params0, class_ids, vc, P, tgt, mask, dict(
                n_iters=n_iters, representation=representation, prior=prior, want_points=want_points,
                want_loss=want_loss, want_traj=want_traj))
        return self._fit_once(params0, class_ids, view_counts, P, tgt, mask, n_iters, representation, prior,
                              want_points, want_loss, want_traj)

    def _fit_grouped(self, params0, class_ids, vc, P, tgt, mask, kw):
        if vc.max() > MAX_VIEWS:
            raise _lib.OdamError(f"views per object must be in 1..{MAX_VIEWS}, got {vc.min()}..{vc.max()}")
        as_t = lambda x: x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
        P, tgt, mask, params0 = as_t(P).reshape(-1, 12), as_t(tgt).reshape(-1, 4), as_t(mask).reshape(-1, 4), as_t(params0).reshape(-1, 9)
        offs = np.concatenate([[0], np.cumsum(vc)])
        need = np.ones(len(vc), np.int64)                      # workgroups an object needs
        big = vc > WG_VIEWS
        need[big] = (2 ** np.ceil(np.log2(vc[big])).astype(np.int64)) // WG_VIEWS
        groups = [np.flatnonzero(~big)] if (~big).any() else []
        for k in sorted(set(need[big].tolist())):
            idx = np.flatnonzero(need == k)
            per = max(8, (self.n_cu // k) // 8 * 8)            # padded object count x k <= number of CUs
            groups += [idx[i:i + per] for i in range(0, len(idx), per)]
        outs = {}
        for g in groups:
            rows = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in g])
            o = self._fit_once(params0[g], [class_ids[i] for i in g], vc[g], P[rows], tgt[rows], mask[rows],
                               kw["n_iters"], kw["representation"], kw["prior"], kw["want_points"], kw["want_loss"], kw["want_traj"])
            for key, val in o.items():
                if val is None:
                    outs[key] = None
                    continue
                if key not in outs:
                    outs[key] = torch.empty((len(vc),) + tuple(val.shape[1:]), device=val.device, dtype=val.dtype)
                outs[key][torch.as_tensor(g, device=val.device)] = val
        return outs

    def _fit_once(self, params0, class_ids, view_counts, P, tgt, mask, n_iters=200, representation="super_quadric",
                  prior=True, want_points=True, want_loss=False, want_traj=False):
        dev = self.device
        n = len(view_counts)
        as_dev = lambda x, dt: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(
            device=dev, dtype=dt).contiguous()
        vc = np.asarray(view_counts, np.int64)
        if n == 0:
            return {"params": torch.zeros(0, 9, device=dev), "points": torch.zeros(0, N_POINTS, 3, device=dev)}
        if vc.min() < 1 or vc.max() > MAX_VIEWS:
            raise _lib.OdamError(f"views per object must be in 1..{MAX_VIEWS}, got {vc.min()}..{vc.max()}")
        offs = np.zeros(n + 1, np.int32)
        offs[1:] = np.cumsum(vc)
        d_p0 = as_dev(params0, torch.float32).reshape(n, 9)
        cls = np.asarray(class_ids, np.int32).copy()
        if prior:
            if cls.min() < 0 or cls.max() > 7:
                raise KeyError(int(cls.max()))  # CLASS_MAPPER covers 0..7 only (sq_libs.py:464)
        else:
            cls[:] = -1
        d_cls = as_dev(cls, torch.int32)
        d_off = as_dev(offs, torch.int32)
        d_P = as_dev(P, torch.float32).reshape(-1, 12)
        d_t = as_dev(tgt, torch.float32).reshape(-1, 4)
        d_m = as_dev(mask, torch.float32).reshape(-1, 4)
        assert d_P.shape[0] == offs[-1] and d_t.shape[0] == offs[-1] and d_m.shape[0] == offs[-1]
        out_p = torch.empty(n, 9, device=dev, dtype=torch.float32)
        out_pts = torch.empty(n, N_POINTS, 3, device=dev, dtype=torch.float32) if want_points else None
        loss = torch.empty(n, n_iters, device=dev, dtype=torch.float32) if want_loss else None
        traj = torch.empty(n, n_iters, 9, device=dev, dtype=torch.float32) if want_traj else None
        with torch.cuda.device(dev), self._lock:
            cur = torch.cuda.current_stream(dev)
            stream = cur.cuda_stream
            if self._last is not None and self._last[0] != stream:
                cur.wait_event(self._last[1])     # the handle's previous launch ran on another stream: order behind it
            _lib.check(_lib.lib().odam_sq_fit_batch(
                self._h, ctypes.c_int(n), _lib.ptr(d_p0), _lib.ptr(d_cls), _lib.ptr(d_off), _lib.ptr(d_P),
                _lib.ptr(d_t), _lib.ptr(d_m), _lib.ptr(self._prior_dev()), ctypes.c_int(int(n_iters)),
                ctypes.c_int(REPRESENTATIONS[representation]), ctypes.c_int(int(vc.max())),
                _lib.ptr(out_p), _lib.ptr(out_pts), _lib.ptr(loss), _lib.ptr(traj),
                ctypes.c_void_p(stream)), "odam_sq_fit_batch")
            ev = torch.cuda.Event()
            ev.record(cur)
            self._last = (stream, ev)
        return {"params": out_p, "points": out_pts, "loss": loss, "traj": traj}

    def points(self, params):
        """compute_ellipsoid_points for [n,9] parameter rows -> [n,1000,3] device tensor."""
        dev = self.device
        d_p = torch.as_tensor(np.asarray(params) if not torch.is_tensor(params) else params).to(
            device=dev, dtype=torch.float32).reshape(-1, 9).contiguous()
        n = d_p.shape[0]
        out = torch.empty(n, N_POINTS, 3, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), self._lock:
            cur = torch.cuda.current_stream(dev)
            stream = cur.cuda_stream
            if self._last is not None and self._last[0] != stream:
                cur.wait_event(self._last[1])
            _lib.check(_lib.lib().odam_sq_points_batch(self._h, ctypes.c_int(n), _lib.ptr(d_p), _lib.ptr(out),
                                                       ctypes.c_void_p(stream)), "odam_sq_points_batch")
            ev = torch.cuda.Event()
            ev.record(cur)
            self._last = (stream, ev)
        return out


def _project_extents(self, params, T_cw, K, on_device=False):
    """[n,9] parameter rows -> [n,4] float64 (x_min, y_min, x_max, y_max) of each surface projected with K @ T_cw[:3]
    (OdamProcess._prepare_tracks, processor.py:181-207), computed on the device"""
    dev = self.device
    d_p = torch.as_tensor(np.asarray(params)).to(device=dev, dtype=torch.float32).reshape(-1, 9).contiguous()
    n = d_p.shape[0]
    out = torch.empty(n, 4, device=dev, dtype=torch.float64)
    cam = np.ascontiguousarray(np.concatenate([np.asarray(T_cw, np.float64)[:3].reshape(-1), np.asarray(K, np.float64)[:3, :3].reshape(-1)]))
    with torch.cuda.device(dev), self._lock:
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        if self._last is not None and self._last[0] != stream:
            cur.wait_event(self._last[1])
        _lib.check(_lib.lib().odam_sq_project_extents(self._h, ctypes.c_int(n), _lib.ptr(d_p), cam.ctypes.data_as(ctypes.c_void_p),
                                                      _lib.ptr(out), ctypes.c_void_p(stream)), "odam_sq_project_extents")
        ev = torch.cuda.Event()
        ev.record(cur)
        self._last = (stream, ev)
    return out if on_device else out.cpu().numpy()


SqFitter.project_extents = _project_extents


def _build_track_windows(self, win, T_cw, K, cam_azi, img_w, img_h):
    """The whole device chain of OdamProcess._prepare_tracks in one native call (include/odam_assoc.h, odam_trackwin_build_tracks):
    parameter rows from the window store's running sums -> surface extents on this fitter -> [T, 79, window] float32 track input
    for the camera T_cw.  `win`: associator.TrackWindows on this fitter's device, in step with the host tracks."""
    dev = self.device
    T = len(win.lengths)
    out = torch.empty(T, 79, win.WINDOW, device=dev, dtype=torch.float32)
    cam = np.ascontiguousarray(np.concatenate([np.asarray(T_cw, np.float64)[:3].reshape(-1), np.asarray(K, np.float64)[:3, :3].reshape(-1)]))
    with torch.cuda.device(dev), self._lock:
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        if self._last is not None and self._last[0] != stream:
            cur.wait_event(self._last[1])
        _lib.check(_lib.lib().odam_trackwin_build_tracks(win._h, self._h, ctypes.c_int(T), cam.ctypes.data_as(ctypes.c_void_p),
                                                         ctypes.c_double(cam_azi), ctypes.c_double(img_w), ctypes.c_double(img_h),
                                                         _lib.ptr(out), ctypes.c_void_p(stream)), "odam_trackwin_build_tracks")
        ev = torch.cuda.Event()
        ev.record(cur)
        self._last = (stream, ev)
    return out


SqFitter.build_track_windows = _build_track_windows


def host_sample(a, e):
    """odam_sq_sample: (a[3], e[2]) -> etas[1000], omegas[1000] float32 (host)."""
    a = np.ascontiguousarray(a, np.float32)
    e = np.ascontiguousarray(e, np.float32)
    et = np.zeros(N_POINTS, np.float32)
    om = np.zeros(N_POINTS, np.float32)
    fp = _lib.c_float_p
    _lib.check(_lib.lib().odam_sq_sample(a.ctypes.data_as(fp), e.ctypes.data_as(fp), et.ctypes.data_as(fp),
                                         om.ctypes.data_as(fp)), "odam_sq_sample")
    return et, om
