"""OdamProcess -- the plugin surface of the reference (likojack/ODAM src/processor.py:15-368) over the
MI355X detector and super-quadric fitter.

Same constructor, attributes (`tracks`, `T_wcs`, `P_cws`, `usable_frames`, `sequence_meta`) and methods
(`init_sequence`, `process_frame`, `run_detector`, `optim_process`, `merge_process`) so that
src/scripts/run_processor.py:61-83 runs unchanged with `from odam_amd.processor import OdamProcess`.

What is different underneath:
  * `detector` is an odam_amd.detector.Detector (gfx950 kernels); `run_detector` can also be fed
    detections computed ahead of time in batches / on other ranks (`detect_frames`, `process_frame(...,
    detections=...)`): detection is stateless per frame (processor.py:259-289), association is not.
  * `_prepare_tracks` evaluates the surface of ALL live tracks in one odam_sq_points_batch launch
    instead of one CPU sampler call per track (processor.py:181-207).
  * `optim_process` batches every object's 200-step fit into one kernel launch (odam_amd.multi_view).
The associator (src/models/associator.py) is the reference's own and stays caller-supplied; the track merge
between the two fit passes is restated in odam_amd/merge.py (run_merge.py:79-130).
"""
import ctypes
import logging

import numpy as np
import torch

from . import merge, multi_view


class _Meta(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def get_homogeneous(pts):  # geometry_utils.py:7-49 (numpy, [N,3] -> [N,4])
    return np.concatenate([pts, np.ones_like(pts[:, 2:])], axis=1)


def projection(pts, intr_mat):  # geometry_utils.py:276-316, unbatched, keep_z=False
    pts = pts @ intr_mat.T
    z = np.ones_like(pts[:, -1])
    pts = pts / pts[:, -1:]
    pts[:, -1] *= z
    return pts


def _track_mark(track, k):
    """what identifies the first k rows of a track for the running sums of _prepare_tracks: frame id and centre of its first
    and its k-th observation (a slot of the track list that now holds another track, or an edited prefix, does not match)"""
    return (track[0, 0], track[0, 9], track[0, 10], track[k - 1, 0], track[k - 1, 9], track[k - 1, 10])


def get_cam_azi(T_wc):  # scannet_utils.py:213-222
    o = np.array([[0, 0, 1], [0, 0, 0]])
    o = (get_homogeneous(o) @ T_wc.T)[:, :3]
    o = o[0] - o[1]
    o[2] = 0
    o = o / np.linalg.norm(o)
    return np.arctan2(o[1], o[0])


def detection_array(out_objects, b, frame_id, seq_w, seq_h):
    """processor.py:269-288 for one frame, all detections at once: [n, 79] float64.  Same arithmetic as the reference's
    per-detection loop (float32 boxes divided in place by the sequence size, float32 angle -> sin / cos, every value
    widened to float64, 64 code columns of -1)."""
    bb = out_objects["bboxes"][b]
    n = len(bb)
    rows = np.full((n, 79), -1.0)
    if n == 0:
        return rows
    angles = np.asarray(out_objects["angles"][b]) / 180. * np.pi
    bb[:, :, 0] /= seq_w
    bb[:, :, 1] /= seq_h
    rows[:, 0] = frame_id
    rows[:, 1] = out_objects["classes"][b]
    rows[:, 2:6] = bb.reshape(n, 4)
    rows[:, 6:9] = out_objects["dimensions"][b]
    rows[:, 9:12] = out_objects["translates"][b]
    rows[:, 12] = np.sin(angles)
    rows[:, 13] = np.cos(angles)
    rows[:, 14] = out_objects["scores"][b]
    return rows


def detection_rows(out_objects, b, frame_id, seq_w, seq_h):
    """processor.py:269-288: one frame's post-processed detections -> list of 79-float rows."""
    return detection_array(out_objects, b, frame_id, seq_w, seq_h).tolist()


class OdamProcess:
    def __init__(self, detector, associator, transforms, scale_prior, detect_threshold=0.6, match_threshold=0.1,
                 score_threshold=0.8, representation="super_quadric", no_code=True, fitter=None):
        self.detector = detector
        self.associator = associator
        self.transforms = transforms
        self.scale_prior = scale_prior
        self.detect_threshold = detect_threshold
        self.match_threshold = match_threshold
        self.score_threshold = score_threshold
        self.representation = representation
        self.no_code = no_code
        self.run_associator = False
        self._tracks = None
        self._pending = []          # fast path: (track ids, [k, 82] rows) attached per frame, not yet concatenated into self._tracks
        self._pending_app = None    # fast path: the last frame's attach, not yet appended to the device mirror (goes with the next step)
        self._n_tracks = 0
        self._exposed = True        # somebody may hold (and edit) the list self.tracks returned: the mirror is re-checked before the next fast frame
        self.sequence_meta = None
        self.fitter = fitter
        self.logger = logging.getLogger('OdamProcess')

    # `tracks` is the reference's attribute (src/processor.py:300; a list of [n, 82] arrays, index = track id).  The fast path of
    # process_frames keeps what a frame attaches as (ids, rows) records and concatenates them into the arrays only when somebody asks
    # for the list -- a concatenate per matched track per frame is a third of that path's host time otherwise.
    @property
    def tracks(self):
        self._sync_host_view()
        self._exposed = True
        return self._tracks

    @tracks.setter
    def tracks(self, value):
        self._pending = []
        self._pending_app = None
        self._tracks = value
        self._n_tracks = len(value) if value is not None else 0
        self._exposed = True

    def _sync_host_view(self):
        """fold the fast path's records into self._tracks and bring the device mirror's host-side bookkeeping (lengths, marks) in step"""
        if self._pending:
            add = {}
            for ids, rows in self._pending:
                for j, t in enumerate(ids):
                    add.setdefault(t, []).append(rows[j:j + 1])
            for t in sorted(add):
                if t < len(self._tracks):
                    self._tracks[t] = np.concatenate([self._tracks[t]] + add[t], axis=0)
                else:
                    assert t == len(self._tracks), (t, len(self._tracks))
                    self._tracks.append(np.concatenate(add[t], axis=0))
            self._pending = []
        win = getattr(self, "_win", None)
        if self._pending_app is not None:       # the last fast frame's attach has not reached the mirror yet
            ids, rows = self._pending_app
            self._pending_app = None
            if win is not None:
                win.append_raw(ids, rows)
        if win is not None and getattr(self, "_win_meta_stale", False):
            win.lengths = [len(t) for t in self._tracks]
            win.marks = win._marks_of(self._tracks)
            self._win_meta_stale = False

    def _fitter(self):
        if self.fitter is None:
            self.fitter = multi_view.default_fitter(str(getattr(self.detector, "device", "cuda:0")))
        return self.fitter

    # ---- sequence state (processor.py:291-304) ---------------------------------------------------
    def init_sequence(self, intrinsics, img_h, img_w):
        self.run_associator = False
        self.sequence_meta = _Meta(K=intrinsics, img_h=img_h, img_w=img_w)
        self.tracks = []
        self._win_meta_stale = False
        self.T_wcs = []
        self.P_cws = []
        self.usable_frames = []
        self._sum_cache = []
        self._win_checked = None
        self._frame_token = None
        if getattr(self, "_win", None) is not None:
            self._win.reset()

    # ---- detection (processor.py:259-289) ----------------------------------------------------------
    def detect_frames(self, rgbs, frame_ids):
        """Batched, stateless run_detector over several frames: list (per frame) of lists of 79-float rows.
        Frames of one size go through the detector in batches of max_batch; frames of DIFFERENT sizes are batched the
        way the reference's DETR.forward batches a list of tensors (padded NestedTensor with a key-padding mask,
        misc.py:303-320) and post-processed each with its own size."""
        size_of = lambda rgb: tuple(rgb.size) if hasattr(rgb, "size") and not isinstance(rgb, np.ndarray) else rgb.shape[1::-1]
        sizes = [size_of(rgb) for rgb in rgbs]
        out = []
        mb = getattr(self.detector, "max_batch", 1)
        on_device = getattr(self.transforms, "on_device", False)
        if on_device:
            self.detector.resize = (self.transforms.size, self.transforms.max_size)
        for b0 in range(0, len(rgbs), mb):
            chunk = rgbs[b0:b0 + mb]
            if len(set(sizes[b0:b0 + mb])) > 1:
                xs = [self.transforms(rgb, None)[0] for rgb in chunk]
                pred = self.detector.forward_nested(xs)
                for b in range(len(chunk)):
                    one = {k: (v[b:b + 1] if torch.is_tensor(v) else v) for k, v in pred.items()}
                    objs = self.detector.postprocess(one, sizes[b0 + b], float(self.detect_threshold), self.sequence_meta.K)
                    out.append(detection_rows(objs, 0, frame_ids[b0 + b], self.sequence_meta.img_w, self.sequence_meta.img_h))
                continue
            if on_device:   # raw uint8 frames up, transform + forward on the GPU
                raw = np.stack([np.asarray(rgb, np.uint8) for rgb in chunk])
                pred = self.detector.forward_u8(torch.from_numpy(raw).to(self.detector.device))
                nb = raw.shape[0]
            else:
                x = torch.stack([self.transforms(rgb, None)[0] for rgb in chunk])
                pred = self.detector(x)
                nb = x.shape[0]
            objs = self.detector.postprocess(pred, sizes[b0], float(self.detect_threshold), self.sequence_meta.K)
            for b in range(nb):
                out.append(detection_rows(objs, b, frame_ids[b0 + b], self.sequence_meta.img_w,
                                          self.sequence_meta.img_h))
        return out

    def run_detector(self, rgb, frame_id, T_wc):
        return self.detect_frames([rgb], [frame_id])[0]

    # ---- tracks (processor.py:95-257) --------------------------------------------------------------
    def _img_size_row(self):
        m = self.sequence_meta
        return np.array([[m.img_w, m.img_h, m.img_w, m.img_h]])

    def _track_rows(self, det, T_wc, with_code):
        """[n,79] detection rows of one frame -> [n,82] track rows (layout processor.py:98-108): boxes in pixels, centre
        and azimuth moved from the camera to the world frame, the detected box repeated as the projected one."""
        rows = np.full((len(det), 82), -1.0)
        px = det[:, 2:6] * self._img_size_row()
        rows[:, :9] = det[:, :9]
        rows[:, 2:6] = px
        rows[:, 9:12] = (get_homogeneous(det[:, 9:12]) @ T_wc.T)[:, :3]
        rows[:, 12] = np.arctan2(det[:, 12], det[:, 13]) + get_cam_azi(T_wc)
        rows[:, 13] = det[:, 14]
        if with_code:
            rows[:, 14:78] = det[:, 15:79]
        rows[:, 78:82] = px
        return rows

    def _init_tracks(self, detections, T_wc):
        """processor.py:95-131: every detection of the first frame with detections starts a track"""
        return [r[None, :] for r in self._track_rows(detections, T_wc, with_code=False)]

    # ---- device-resident track windows (odam_amd.associator.TrackWindows) ------------------------------------------------
    def _windows(self):
        """the device mirror of self.tracks' observations, or None where it does not apply (a caller-supplied associator
        or fitter without the native entry points, shape codes in use, more tracks than the native store holds)"""
        self._win_checked = None      # set again below only when the mirror was checked against self.tracks in THIS frame
        if not (getattr(self.associator, "native_windows", False) and self.no_code and hasattr(self._fitter(), "project_extents")):
            return None
        if getattr(self, "_win", None) is None:
            from .associator import TrackWindows
            self._win = TrackWindows(str(getattr(self.associator, "device", "cuda:0")))
        w = self._win
        if len(self.tracks) > w.max_tracks:
            return None
        # the window kernel reads the fitter's projected extents through a raw pointer on the window store's stream: both must
        # live on one device (a detector + fitter on cuda:0 with the associator on cuda:1 takes the host path)
        fdev = getattr(self._fitter(), "device", None)
        if fdev is not None and torch.device(fdev) != w.device:
            return None
        if not w.in_step(self.tracks):                        # first use, or somebody edited self.tracks: rebuild the mirror
            w.load(self.tracks)
        self._win_checked = getattr(self, "_frame_token", None)   # _attach_to_tracks of this frame need not check again
        return w

    def _preprocess_tracks(self, tracks, T_wc, cam_azi, n_times=100, proj_px=None):
        """processor.py:133-170: the last `n_times` observations of every track in the CURRENT camera frame,
        [T, n_times, 79] float32 padded with -1.  All observations of all tracks go through numpy at once (the
        reference loops over tracks; every row is computed by the same operations), and nothing here is a torch CPU
        op: on a many-core host those fork the whole OpenMP pool, which costs milliseconds per frame in this loop."""
        in_ = np.full((len(tracks), n_times, 79), -1.0, np.float32)
        if not len(tracks):
            return in_
        tails = [t[-n_times:] for t in tracks]                       # only the window that is used (rows are independent)
        lens = np.array([len(t) for t in tails])
        ends = np.cumsum(lens)
        cat = np.concatenate(tails, axis=0)                          # [sum min(n, n_times), 82]
        # the box projected by _prepare_tracks: handed over directly, or (reference calling convention) in the last row
        proj = cat[ends - 1, -4:] if proj_px is None else proj_px
        assert not np.any(np.all(proj == -1, axis=1)), "wrong projected bbox"
        proj = np.clip(proj / self._img_size_row()[0], a_min=-1, a_max=2)
        obs = np.full((len(cat), 79), -1.0)
        obs[:, :9] = cat[:, :9]
        obs[:, 2:6] = proj[np.repeat(np.arange(len(tracks)), lens)]
        obs[:, 9:12] = (get_homogeneous(cat[:, 9:12]) @ np.linalg.inv(T_wc).T)[:, :3]
        rel = cat[:, 12] - cam_azi
        obs[:, 12], obs[:, 13] = np.sin(rel), np.cos(rel)
        obs[:, 14] = cat[:, 13]
        obs[:, 15:79] = cat[:, 14:78]
        obs = obs.astype(np.float32)                                 # torch.tensor(., dtype=float32): round to nearest
        for idx, (n, e) in enumerate(zip(lens, ends)):
            k = min(int(n), n_times)
            in_[idx, :k] = obs[e - k:e]
        return in_

    def _prepare_tracks(self, T_wc, n_times=100):
        cam_azi = get_cam_azi(T_wc)
        tracks = self.tracks                         # read-only here (the reference deep-copies and scribbles on the copy)
        fitter = self._fitter()
        win = self._windows() if len(tracks) and n_times == getattr(getattr(self, "_win", None), "WINDOW", 100) else None
        if win is not None and hasattr(fitter, "build_track_windows"):
            # the device mirror is in step with `tracks` (checked by _windows()): the per-track means below come from its running
            # sums, and surface extents + window tensor follow in the same native call (odam_trackwin_build_tracks) -- nothing
            # of this frame's track input is computed or touched on the host
            m = self.sequence_meta
            return fitter.build_track_windows(win, np.linalg.inv(T_wc), m.K, cam_azi, m.img_w, m.img_h)
        # one batched surface evaluation for all tracks (reference: one SuperQuadric + sampler call each);
        # parameters as sq.init_params builds them: mean centre, mean azimuth, sqrt(mean dims / 2), shapes -0
        params = np.zeros((len(tracks), 9))
        params[:, 7:] = -0.0
        # np.mean = np.add.reduce / n.  Over axis 0 of an [n, 3] block numpy adds the rows one after the other, so a running sum
        # per track (extended by the rows a frame appends) gives the same bits as reducing the whole track again; the 1-D
        # azimuth column is reduced pairwise by numpy and is therefore reduced afresh.
        cache = getattr(self, "_sum_cache", None)
        if cache is None:
            cache = self._sum_cache = []
        del cache[len(tracks):]
        for i, track in enumerate(tracks):
            n = len(track)
            if i < len(cache) and cache[i][0] == n and cache[i][3] == _track_mark(track, n):
                _, s_t, s_d = cache[i][:3]
            elif i < len(cache) and 0 < cache[i][0] < n and cache[i][3] == _track_mark(track, cache[i][0]):
                k, s_t, s_d = cache[i][:3]
                s_t = s_t.copy(); s_d = s_d.copy()
                for r in range(k, n):
                    s_t += track[r, 9:12]; s_d += track[r, 6:9]
                cache[i] = (n, s_t, s_d, _track_mark(track, n))
            else:
                s_t = np.add.reduce(track[:, 9:12], axis=0); s_d = np.add.reduce(track[:, 6:9], axis=0)
                entry = (n, s_t, s_d, _track_mark(track, n))
                if i < len(cache):
                    cache[i] = entry
                else:
                    cache.append(entry)
            params[i, :3] = s_t / n
            params[i, 3] = np.add.reduce(track[:, 12], axis=0) / n
            params[i, 4:7] = np.maximum(s_d / n, 0.05)    # np.clip(., 0.05, inf)
        params[:, 4:7] = np.sqrt(params[:, 4:7] / 2)
        params = params.astype(np.float32)
        if win is not None:
            # everything after the per-track means on the device: surface extents (odam_sq_project_extents), then the
            # [T, 79, n_times] window tensor in the current camera frame (odam_trackwin_build) -- nothing comes back to the host
            T_cw = np.linalg.inv(T_wc)
            proj_dev = fitter.project_extents(params, T_cw, self.sequence_meta.K, on_device=True)
            return win.build(proj_dev, T_cw, cam_azi, self.sequence_meta.img_w, self.sequence_meta.img_h)
        if len(tracks) and hasattr(fitter, "project_extents"):
            # surface points, camera transform, projection and min / max on the device (odam_sq_project_extents): the same
            # float64 products per point as below; 32 bytes per track come back
            proj_px = fitter.project_extents(params, np.linalg.inv(T_wc), self.sequence_meta.K)
        elif len(tracks):
            # (a fitter without the device projection -- the oracle-backed test double: tests/test_processor.py)
            pts_all = fitter.points(params).cpu().numpy()
            T_cw = np.linalg.inv(T_wc)
            cam = (np.concatenate([pts_all, np.ones_like(pts_all[..., 2:])], axis=2) @ T_cw.T)[..., :3]
            pix = cam @ np.asarray(self.sequence_meta.K).T
            pix = np.ascontiguousarray((pix / pix[..., -1:]).transpose(0, 2, 1))      # [T, 3, 1000]
            lo, hi = pix.min(axis=2), pix.max(axis=2)
            proj_px = np.concatenate([lo[:, :2], hi[:, :2]], axis=1)                  # x_min, y_min, x_max, y_max
        else:
            proj_px = np.zeros((0, 4))
        in_ = self._preprocess_tracks(tracks, T_wc, cam_azi, n_times, proj_px=proj_px)
        return torch.from_numpy(np.ascontiguousarray(in_.transpose(0, 2, 1)))   # [T, 79, n_times], as .permute(0, 2, 1)

    def _collater(self, data_list):
        max_dets = 30
        n_features = data_list[0]['detections'].shape[0]
        detections = np.full((len(data_list), n_features, max_dets), -1.0, np.float32)
        tracks, poses, tsplit, dsplit, valid = [], [], [], [], []
        for b_id, data in enumerate(data_list):
            tracks.append(data['tracks'])
            poses.append(data['pose'])
            nt, nd = data['tracks'].shape[0], data['detections'].shape[1]
            tsplit.append(nt); dsplit.append(nd); valid.append((nt, nd))
            detections[b_id, :, :nd] = np.asarray(data['detections'], np.float32)
        gt_masks = np.zeros((sum(tsplit), sum(dsplit)), np.float32)
        t0 = d0 = 0
        for nt, nd in zip(tsplit, dsplit):
            gt_masks[t0:t0 + nt, d0:d0 + nd] = 1
            t0 += nt; d0 += nd
        return {"tracks": tracks[0] if len(tracks) == 1 else torch.cat(tracks, dim=0),
                "detections": torch.from_numpy(detections), "gt_masks": torch.from_numpy(gt_masks),
                "track_batch_split": tsplit, "detection_batch_split": dsplit, "poses": poses, "valid_list": valid}

    def _attach_to_tracks(self, pred_match, detections, T_wc, score_mat):
        """processor.py:212-257: a matched detection extends its track, an unmatched one (-1) starts a new one;
        matches scoring below `score_threshold` are dropped (index -1 reads the dustbin row, as in the reference)."""
        rows = self._track_rows(detections, T_wc, with_code=not self.no_code)
        win = getattr(self, "_win", None)
        # the device mirror matches the tracks as they stand: checked by _windows() in this very frame, or checked now
        tok = getattr(self, "_frame_token", None)
        in_step = win is not None and ((tok is not None and getattr(self, "_win_checked", None) is tok) or win.in_step(self.tracks))
        ids, picked = [], []
        for det_id, track_id in enumerate(pred_match):
            if score_mat[track_id, det_id] < self.score_threshold:
                continue
            if track_id == -1:
                ids.append(len(self.tracks))
                self.tracks.append(rows[det_id:det_id + 1])
            else:
                ids.append(int(track_id))
                self.tracks[track_id] = np.concatenate([self.tracks[track_id], rows[det_id:det_id + 1]], axis=0)
            picked.append(det_id)
        if in_step and ids and len(set(ids)) == len(ids) and max(ids) < win.max_tracks:
            win.append(ids, rows[picked])       # keep the mirror in step (whenever it is not, _windows() rebuilds it from self.tracks)

    # ---- per-frame driver (processor.py:306-345) ---------------------------------------------------
    def process_frame(self, rgb, frame_id, T_wc, detections=None):
        """`detections`: optional rows for this frame computed earlier by `detect_frames` (here or on another
        rank); None -> run the detector now, as the reference does."""
        if detections is not None and self._fast_ok():
            return self.process_frames([frame_id], [T_wc], [detections])
        return self._process_frame_host(rgb, frame_id, T_wc, detections)

    def _process_frame_host(self, rgb, frame_id, T_wc, detections=None):
        """process_frame with every step a separate call, as the reference has them (src/processor.py:306-345): what runs when the
        associator or the fitter is the caller's own (no native track store), with shape codes, or with the detector in the loop"""
        self._frame_token = object()      # one per call: what "_windows() checked the device mirror in this very frame" is keyed on
        self.usable_frames.append(frame_id)
        self.T_wcs.append(T_wc)
        self.P_cws.append(self.sequence_meta.K @ np.linalg.inv(T_wc)[:3, :])
        if detections is None:
            detections = self.run_detector(rgb, frame_id, T_wc)
        if len(detections) == 0:
            return None
        detections = np.asarray(detections)
        if len(detections) > 30:
            detections = detections[:30, :]
        if not self.run_associator:
            self.run_associator = True
            self.tracks = self._init_tracks(detections, T_wc)
            return None
        dev = getattr(self.detector, "device", "cuda")
        track_tensors = self._prepare_tracks(T_wc)
        data = self._collater([{"detections": detections.astype(np.float32).T, "tracks": track_tensors, "pose": T_wc}])
        n_tracks, n_detections = data['valid_list'][0]
        with torch.no_grad():
            predictions = self.associator(data, self.match_threshold, eval_only=True)
        pred_matches = predictions['matches'][0].astype(np.int32)
        score_mat = predictions['pred'][0][0].cpu().exp().numpy()
        dets = data['detections'][0, :, :n_detections].numpy().T
        self._attach_to_tracks(pred_matches, dets, T_wc, score_mat)

    # ---- the same loop for frames whose detections are at hand, with the host out of the device's way -------------------------------
    def _fast_ok(self):
        """process_frames' fast path applies: the associator and the fitter are this package's (native track store, odam_assoc_step),
        no shape codes, one device, the Hungarian step on the host (scipy, the default)"""
        a = self.associator
        if not (self.no_code and getattr(a, "native_windows", False) and hasattr(a, "step_launch") and self._tracks is not None):
            return False      # (decided before the fitter is touched: a caller with its own associator may have no GPU fitter at all)
        if "assignment" in vars(a) or "_attach_to_tracks" in vars(self) or "_prepare_tracks" in vars(self):
            return False      # somebody hooked a step of the per-frame path on this object (tests do): keep the steps separate
        f = self._fitter()
        if not (hasattr(f, "build_track_windows") and hasattr(f, "_h")):
            return False
        from . import _lib
        if _lib.get_config("assoc.hungarian"):
            return False
        adev, fdev = getattr(a, "device", None), getattr(f, "device", None)
        return adev is not None and fdev is not None and torch.device(adev) == torch.device(fdev)

    def process_frames(self, frame_ids, T_wcs, detections_list):
        """process_frame(None, frame_ids[k], T_wcs[k], detections=detections_list[k]) for k = 0, 1, ... -- the loop of
        run_processor.py:70-79 for frames whose detections were computed ahead (detect_frames, another rank's all-gather) -- and the same
        results: the same library calls on the same values in the same order.  What changes is where the host's work sits.  Per frame the
        device runs one chain (mirror append -> track input -> association network -> Sinkhorn, ~0.47 ms at 40 tracks) and the host has
        ~0.25 ms of its own (camera quantities, track rows, Hungarian step, list bookkeeping); frame by frame the two alternate, each
        idle while the other works.  Here a frame's device work is ONE native call (odam_assoc_step), everything the host can know
        before the assignment arrives -- the NEXT frame's camera quantities and track rows included -- is computed while the device works,
        and the track list is extended lazily (self.tracks folds the records in when it is read).  Only the Hungarian step and the
        threshold tests stay between one frame's result and the next frame's launch."""
        n = len(frame_ids)
        if not self._fast_ok():
            for k in range(n):
                self._process_frame_host(None, frame_ids[k], T_wcs[k], detections_list[k])
            return None
        m = self.sequence_meta
        K = np.asarray(m.K, np.float64)
        K9 = K[:3, :3].reshape(-1)
        assoc, fitter = self.associator, self._fitter()

        def prep(k):
            """everything of frame k the host can compute without the association result (numpy, as the per-frame path computes it)"""
            T_wc = T_wcs[k]
            det = detections_list[k]
            T_cw = np.linalg.inv(T_wc)
            P = {"T_wc": T_wc, "P_cw": m.K @ T_cw[:3, :], "n": len(det)}
            if len(det) == 0:
                return P
            det = np.asarray(det)
            if len(det) > 30:
                det = det[:30, :]
            # the float32 round trip of the collater (processor.py:49-68, 337-339): the attach reads the detections back from the
            # float32 tensor the associator was given
            det32T = det.astype(np.float32).T
            P["det32T"] = det32T
            P["det"] = det
            P["rows82"] = self._track_rows(det32T.T, T_wc, with_code=False)
            P["cam_azi"] = get_cam_azi(T_wc)
            P["cam21"] = np.ascontiguousarray(np.concatenate([T_cw[:3].reshape(-1), K9]))
            return P

        nxt = prep(0) if n else None
        for k in range(n):
            cur, nxt = nxt, None
            self.usable_frames.append(frame_ids[k])
            self.T_wcs.append(cur["T_wc"])
            self.P_cws.append(cur["P_cw"])
            launched = False
            if cur["n"]:
                if not self.run_associator:
                    # the first frame with detections starts the tracks (processor.py:320-323): host rows, the mirror is loaded below
                    self.run_associator = True
                    self.tracks = self._init_tracks(cur["det"], cur["T_wc"])
                else:
                    win = self._fast_mirror()
                    if win is None:      # more tracks than the native store holds: this frame takes the host path
                        self._rewind_frame()
                        self._process_frame_host(None, frame_ids[k], T_wcs[k], detections_list[k])
                        if k + 1 < n:
                            nxt = prep(k + 1)
                        continue
                    app = self._pending_app or ((), np.zeros((0, 82)))
                    self._pending_app = None
                    try:
                        assoc.step_launch(win, fitter, app[0], np.asarray(app[1])[:, :14], self._n_tracks, cur["cam21"], float(cur["cam_azi"]),
                                          float(m.img_w), float(m.img_h), cur["det32T"])
                    except BaseException:
                        self._mirror_unknown()
                        raise
                    launched = True
            if k + 1 < n:
                nxt = prep(k + 1)                 # under the device's work on frame k
            if launched:
                try:
                    Z = assoc.step_finish()
                except BaseException:
                    self._mirror_unknown()
                    raise
                self._attach_fast(Z, cur)
        return None

    def _mirror_unknown(self):
        """a native call of the fast path failed: what the device mirror holds is not known any more (the append of the last attach may or may
        not have run) -- forget its host-side bookkeeping, so that whoever uses it next (either path) rebuilds it from the track list"""
        self._pending_app = None
        self._exposed = True
        self._win_meta_stale = False
        win = getattr(self, "_win", None)
        if win is not None:
            win.lengths = [-1]          # no track list has this: in_step() fails, load() follows

    def _rewind_frame(self):
        self.usable_frames.pop(); self.T_wcs.pop(); self.P_cws.pop()

    def _fast_mirror(self):
        """the device mirror, in step with the tracks as the fast path knows them; None where the native store does not apply"""
        if getattr(self, "_win", None) is None:
            from .associator import TrackWindows
            self._win = TrackWindows(str(getattr(self.associator, "device", "cuda:0")))
        w = self._win
        if self._exposed:
            # the list was handed out (or replaced) since the last fast frame: fold everything in and check the mirror against it once
            self._sync_host_view()
            self._n_tracks = len(self._tracks)
            if self._n_tracks + 30 > w.max_tracks:
                return None
            if not w.in_step(self._tracks):
                w.load(self._tracks)
            self._exposed = False
        elif self._n_tracks + 30 > w.max_tracks:
            return None
        return w

    def _attach_fast(self, Z, cur):
        """hungarian_matching + _attach_to_tracks (src/models/associator.py:19-35, src/processor.py:212-257) on the assignment of a
        fast frame: the same tests on the same float32 values; the rows go to the record list and to the mirror's next append"""
        from . import _lib
        n_det = cur["det32T"].shape[1]
        # hungarian_matching (associator.py:19-35) on the float32 scores torch computes -- linear_sum_assignment(1 - scores), kept where the
        # matched score exceeds the match threshold -- and _attach_to_tracks' test (processor.py:228-231: a detection is dropped if the score
        # of its pair, for an unmatched one the dustbin row, is below the score threshold; a NaN is not below anything: kept, as in the
        # loop): ONE native host call on ONE exponential (odam_assoc_attach restates scipy's solver with its tie order;
        # tests/test_assoc_host.py holds it to scipy on random, tie-ridden and fixture matrices)
        score_mat = Z.exp()
        att = getattr(self, "_att", None)
        if att is None:      # result buffers and their addresses, once (numpy's .ctypes.data_as costs 3 us a call, np.empty + two of them 7)
            m_, k_ = np.empty(32, np.int32), np.empty(32, np.uint8)
            att = self._att = (m_, k_, ctypes.c_void_p(m_.ctypes.data), ctypes.c_void_p(k_.ctypes.data), _lib.lib().odam_assoc_attach)
        rc = att[4](ctypes.c_void_p(score_mat.data_ptr()), Z.shape[0] - 1, n_det, Z.shape[1], ctypes.c_double(self.match_threshold),
                    ctypes.c_double(self.score_threshold), att[2], att[3])
        if rc in (4, 5):      # what scipy raises for the same matrices
            raise ValueError("matrix contains invalid numeric entries" if rc == 4 else "cost matrix is infeasible")
        _lib.check(rc, "odam_assoc_attach")
        match, keep = att[0][:n_det], att[1][:n_det]
        picked = np.flatnonzero(keep)
        mk = match[picked]
        new = mk == -1
        n_tracks = self._n_tracks
        ids_a = mk.astype(np.int64)
        ids_a[new] = n_tracks + np.arange(int(new.sum()))
        n_tracks += int(new.sum())
        ids = ids_a.tolist()
        rows = cur["rows82"]
        if ids:
            assert len(set(ids)) == len(ids), ids      # a track receives at most one observation per frame (the Hungarian step's matching)
            sel = rows[picked]
            self._pending.append((ids, sel))
            self._pending_app = (ids, sel)
            self._n_tracks = n_tracks
            self._win_meta_stale = True

    # ---- back end (processor.py:347-368) -----------------------------------------------------------
    def merge_process(self, data):
        self.logger.info("Merging tracks")
        return merge.merge_process(data, self.usable_frames)

    def optim_process(self, tracks, return_params=False):
        m = self.sequence_meta
        return multi_view.optim_process(tracks, self.usable_frames, self.T_wcs, self.P_cws, m.img_h, m.img_w, m.K,
                                        self.representation, prior=True, n_iters=200, n_views=10,
                                        fitter=self._fitter(), return_params=return_params)

    def optim_process_params(self, tracks):
        """optim_process + the fitted parameter rows and the fitted / not-fitted flags (what a rank sends to the others:
        odam_amd/pipeline.py)"""
        return self.optim_process(tracks, return_params=True)
