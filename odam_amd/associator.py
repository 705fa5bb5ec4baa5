"""Host side of the association network: drop-in for the reference's Associator object as OdamProcess calls it
(likojack/ODAM src/models/associator.py:163-268, src/processor.py:327-341):
    out = associator(data, match_threshold, eval_only=True)
    out["matches"][0]  -> track id per detection (-1 = new track)      (hungarian_matching, :19-35)
    out["pred"][0][0]  -> log assignment [(T+1), (n_det+1)] tensor      (log_optimal_transport, :283-298)
The network forward and the 100 Sinkhorn iterations run in the gfx950 library (include/odam_assoc.h).  The Hungarian step is scipy's on
the host, exactly as the reference does (odam_config assoc.hungarian = 0, the default), or the library's (assoc.hungarian = 1:
odam_assoc_hungarian restates scipy's shortest-augmenting-path solver with its tie order on one wavefront, launched behind the Sinkhorn
loop; same matching on every fixture and on matrices full of ties -- tests/test_assoc_gpu.py -- but 0.01 ms per frame slower, because the
host has to wait for the device either way)."""
import contextlib
import ctypes
import math

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import _lib


def hungarian_matching(scores, matching_threshold):  # associator.py:19-35
    scores = scores.detach().cpu().numpy()
    match = np.zeros(scores.shape[1]) - 1
    rows, cols = linear_sum_assignment(1 - scores)
    for r, c in zip(rows, cols):
        if scores[r, c] > matching_threshold:
            match[c] = r
    return match


class TrackWindows:
    """Device-resident observations of the live tracks (include/odam_assoc.h, odam_trackwin_*): what
    OdamProcess._preprocess_tracks (src/processor.py:133-170) rebuilds on the host every frame is built by one launch here."""

    WINDOW = 100

    def __init__(self, device="cuda:0", max_tracks=1024):
        self.device = torch.device(device)
        self.max_tracks = max_tracks
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_trackwin_create(max_tracks, self.WINDOW, ctypes.byref(self._h)), "odam_trackwin_create")
        self.lengths = []           # observations per track, as appended (the host's check against its own track list)
        self.marks = np.zeros((0, 3))      # per track: frame id of its first observation, frame id and world x of its last -- see in_step()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_trackwin_reset(self._h, self._stream()), "odam_trackwin_reset")
        self.lengths = []
        self.marks = np.zeros((0, 3))

    @staticmethod
    def _marks_of(tracks):
        """[T, 3] float64: frame id of the first observation, frame id and world x of the last one (one pass over the list)"""
        n = len(tracks)
        out = np.empty((n, 3))
        out[:, 0] = np.fromiter((t[0, 0] for t in tracks), np.float64, n)
        out[:, 1] = np.fromiter((t[-1, 0] for t in tracks), np.float64, n)
        out[:, 2] = np.fromiter((t[-1, 9] for t in tracks), np.float64, n)
        return out

    def in_step(self, tracks):
        """True when the mirror holds exactly `tracks` as far as the host can tell without reading rows back: the same number of
        observations per track AND the same first / last observation (frame ids, world x of the last one) -- a track list that was
        replaced or edited at equal lengths (a restore, a test) is caught as well."""
        if [len(t) for t in tracks] != self.lengths:
            return False
        return len(tracks) == 0 or np.array_equal(self._marks_of(tracks), self.marks[:len(tracks)])

    def load(self, tracks):
        """(re)build the mirror from a host track list in one upload (include/odam_assoc.h, odam_trackwin_load)"""
        T = len(tracks)
        if T > self.max_tracks:
            raise _lib.OdamError(f"TrackWindows: {T} tracks exceed the store's {self.max_tracks}")
        lens = np.ascontiguousarray([len(t) for t in tracks], np.int32)
        tails = [np.asarray(t, np.float64)[-self.WINDOW:, :14] for t in tracks if len(t)]
        rows = np.ascontiguousarray(np.concatenate(tails, axis=0)) if tails else np.zeros((0, 14))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_trackwin_load(self._h, ctypes.c_int(T), lens.ctypes.data_as(ctypes.c_void_p),
                                                     rows.ctypes.data_as(ctypes.c_void_p), self._stream()), "odam_trackwin_load")
        self.lengths = lens.tolist()
        if any(n == 0 for n in self.lengths):
            raise _lib.OdamError("TrackWindows: a track without observations")
        self.marks = self._marks_of(tracks)
        # ... and the store's running sums (odam_trackwin_load_sums): dims, centre and azimuth of EVERY observation
        cols = np.ascontiguousarray(np.concatenate([np.asarray(t, np.float64)[:, 6:13] for t in tracks], axis=0)) if T else np.zeros((0, 7))
        _lib.check(_lib.lib().odam_trackwin_load_sums(self._h, ctypes.c_int(T), lens.ctypes.data_as(ctypes.c_void_p),
                                                      cols.ctypes.data_as(ctypes.c_void_p)), "odam_trackwin_load_sums")

    def append(self, track_ids, rows82):
        """rows82 [n, 82] float64 track rows (processor.py:98-108); track_ids [n]: existing index or len(lengths) + j for new ones"""
        rows = np.ascontiguousarray(np.asarray(rows82, np.float64)[:, :14])
        ids = np.ascontiguousarray(track_ids, np.int32)
        for i0 in range(0, len(ids), 32):
            n = min(32, len(ids) - i0)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().odam_trackwin_append(self._h, ctypes.c_int(n), ids[i0:].ctypes.data_as(ctypes.c_void_p),
                                                           rows[i0:].ctypes.data_as(ctypes.c_void_p), self._stream()), "odam_trackwin_append")
        r = np.asarray(rows82, np.float64)
        n_new = int((ids >= len(self.lengths)).sum())
        if n_new:
            self.marks = np.concatenate([self.marks, np.zeros((n_new, 3))])
        for t, row in zip(ids.tolist(), r):
            if t == len(self.lengths):
                self.lengths.append(1)
                self.marks[t, 0] = row[0]
            else:
                self.lengths[t] += 1
            self.marks[t, 1] = row[0]; self.marks[t, 2] = row[9]

    def append_raw(self, track_ids, rows82):
        """append() without the host-side lengths / marks: OdamProcess's fast path keeps those stale while it runs and rebuilds them from
        its track list when somebody looks (processor.py::_sync_host_view)"""
        rows = np.ascontiguousarray(np.asarray(rows82, np.float64)[:, :14])
        ids = np.ascontiguousarray(track_ids, np.int32)
        for i0 in range(0, len(ids), 32):
            n = min(32, len(ids) - i0)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().odam_trackwin_append(self._h, ctypes.c_int(n), ids[i0:].ctypes.data_as(ctypes.c_void_p),
                                                           rows[i0:].ctypes.data_as(ctypes.c_void_p), self._stream()), "odam_trackwin_append")

    def build(self, proj_px_dev, T_cw, cam_azi, img_w, img_h):
        T = len(self.lengths)
        out = torch.empty(T, 79, self.WINDOW, device=self.device, dtype=torch.float32)
        cam = np.ascontiguousarray(np.concatenate([np.asarray(T_cw, np.float64)[:3].reshape(-1), [cam_azi, img_w, img_h]]))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_trackwin_build(self._h, ctypes.c_int(T), _lib.ptr(proj_px_dev), cam.ctypes.data_as(ctypes.c_void_p),
                                                      _lib.ptr(out), self._stream()), "odam_trackwin_build")
        return out

    def params(self):
        """[T, 9] float32 parameter rows of the track surfaces from the store's running sums (odam_trackwin_params) -- what
        OdamProcess._prepare_tracks computes from the host track list (src/processor.py:172-180), bit for bit"""
        T = len(self.lengths)
        out = np.zeros((T, 9), np.float32)
        _lib.check(_lib.lib().odam_trackwin_params(self._h, ctypes.c_int(T), out.ctypes.data_as(ctypes.c_void_p)), "odam_trackwin_params")
        return out

    def close(self):
        if self._h:
            _lib.lib().odam_trackwin_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Associator:
    native_windows = True       # OdamProcess may hand this associator a device tensor built by TrackWindows

    def __init__(self, config=None, max_tracks=256, device="cuda:0"):
        cfg = config or {"GNN_layers": ["self", "cross"] * 6, "self_GNN_layers": ["self"] * 4, "sinkhorn_iterations": 100}
        self.gnn_layers = list(cfg["GNN_layers"])
        self.n_self = len(cfg["self_GNN_layers"])
        self.iters = int(cfg["sinkhorn_iterations"])
        self.max_tracks = max_tracks
        self.device = torch.device(device)
        self._h = None
        self._sd = None
        self.lost_launches = 0      # frames whose persistent matching launch was abandoned and re-run (see assignment_checked)
        self._lost_seen = 0

    def to(self, device):
        self.device = torch.device("cuda:0" if str(device) == "cuda" else device)
        return self

    def cuda(self):
        return self.to("cuda:0")

    def eval(self):
        return self

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in sd.items() if torch.is_tensor(v)}
        self.close()
        return self

    def close(self):
        if self._h:
            _lib.lib().odam_assoc_destroy(self._h)
            self._h = None
            self._lost_seen = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if self._h:
            return self._h
        if self._sd is None:
            raise _lib.OdamError("Associator: load_state_dict() first")
        L = _lib.lib()
        h = ctypes.c_void_p()
        cross = (ctypes.c_int * len(self.gnn_layers))(*[1 if n == "cross" else 0 for n in self.gnn_layers])
        with torch.cuda.device(self.device):
            _lib.check(L.odam_assoc_create(self.max_tracks, self.n_self, cross, len(self.gnn_layers), self.iters,
                                           ctypes.byref(h)), "odam_assoc_create")
            try:
                tensors = dict(self._sd)
                # PositionalEncoding.div_term (associator.py:320), torch CPU float32 ops
                tensors["pe_div_term"] = torch.exp(torch.arange(0, 256, 2).float() * (-math.log(10000.0) / 256))
                for name, t in tensors.items():
                    t = t.reshape(t.shape[0], t.shape[1]) if t.dim() == 3 else t     # Conv1d weight [N, K, 1]
                    t = t.contiguous()
                    shape = (ctypes.c_longlong * max(t.dim(), 1))(*t.shape)
                    _lib.check(L.odam_assoc_set_weight(h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()),
                               f"odam_assoc_set_weight({name})")
                _lib.check(L.odam_assoc_finalize(h), "odam_assoc_finalize")
            except Exception:
                L.odam_assoc_destroy(h)      # a half-built handle is not kept
                raise
        self._h = h
        return h

    MAX_TRACKS = 1024      # odam_assoc_create's limit (the Sinkhorn kernel keeps the whole matrix in LDS)

    def _grow(self, n_tracks):
        """the reference has no track limit (associator.py:202-268): the native workspace is sized per handle, so a scene
        with more live tracks gets a new handle with at least twice the room (weights are kept in self._sd)"""
        if n_tracks > self.MAX_TRACKS:
            raise _lib.OdamError(f"Associator: {n_tracks} live tracks exceed the native limit of {self.MAX_TRACKS}")
        old, old_h = self.max_tracks, self._h
        self.max_tracks = min(self.MAX_TRACKS, max(n_tracks, 2 * self.max_tracks))
        torch.cuda.synchronize(self.device)
        self._h = None
        try:
            self._handle()
        except Exception:
            self.max_tracks, self._h = old, old_h      # the old handle stays usable for smaller frames
            raise
        if old_h:
            _lib.lib().odam_assoc_destroy(old_h)
        self._lost_seen = 0

    def set_persistent(self, on):
        """matching GNN as one persistent launch (the default where the device can hold its 256 workgroups at once) or as the launch
        sequence.  A caller that runs the association on a stream restricted to a share of the compute units
        (hipExtStreamCreateWithCUMask; tests/native/cumask_probe.hip) should switch the persistent launch off: its workgroups wait for each other and would
        never all become resident (the bounded wait would abandon every launch and re-run it through the sequence)."""
        _lib.check(_lib.lib().odam_assoc_set_persistent(self._handle(), ctypes.c_int(1 if on else 0)), "odam_assoc_set_persistent")

    def assignment(self, tracks, detections, n_tracks, n_det, sequence=False, threshold=None):
        """tracks [T,79,100], detections [1,79,30] or [79,30] (any device) -> Z [(T+1),(n_det+1)] device tensor.
        Stream-ordered; sequence=True takes the launch-sequence path (no device-wide barrier).

        This is the RAW entry: an abandoned persistent launch (include/odam_assoc.h) leaves Z all NaN and is reported only
        through odam_assoc_lost_launches -- use assignment_checked() (what __call__ does) unless the caller synchronises and
        checks that counter itself, as the timing scripts do.  A loss taken here is attributed to the next checked call."""
        dev = self.device
        if n_tracks > self.max_tracks:
            self._grow(n_tracks)
        tr = tracks.to(dev, torch.float32).contiguous()
        _pinned = getattr(self, "_pin_mode", False) and (not detections.is_cuda) and detections.dtype == torch.float32
        if _pinned:
            # (inside assignment_checked only.)  The kernels read the detections from, and write Z to, pinned host memory themselves --
            # no copy command either way; this call then waits for the stream itself, so what it returns is valid host data and the
            # two buffers are free for the next frame
            if getattr(self, "_pin", None) is None:
                self._pin = (torch.empty(79 * 30, dtype=torch.float32).pin_memory(), torch.empty((self.MAX_TRACKS + 1) * 31, dtype=torch.float32).pin_memory())
            de = self._pin[0].view(79, 30)
            de.copy_(detections.reshape(79, 30))
            Z = self._pin[1][:(n_tracks + 1) * (n_det + 1)].view(n_tracks + 1, n_det + 1)
        else:
            de = detections.to(dev, torch.float32).reshape(79, 30).contiguous()
            Z = torch.empty(n_tracks + 1, n_det + 1, device=dev, dtype=torch.float32)
        L = _lib.lib()
        fn, what = (L.odam_assoc_forward_sequence, "odam_assoc_forward_sequence") if sequence else (L.odam_assoc_forward, "odam_assoc_forward")
        with torch.cuda.device(dev):
            _lib.check(fn(self._handle(), _lib.ptr(tr), ctypes.c_int(n_tracks), _lib.ptr(de), ctypes.c_int(n_det), _lib.ptr(Z),
                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), what)
            self._match = None
            if threshold is not None and _lib.get_config("assoc.hungarian") and min(n_tracks, n_det) <= 32 and max(n_tracks, n_det) <= 128 \
                    and n_det <= 31:      # the pinned result is 31 match words + the status word
                # the Hungarian step behind the Sinkhorn loop on the same stream: match [n_det] + status into pinned memory next to Z
                if getattr(self, "_pin_match", None) is None:
                    self._pin_match = torch.empty(32, dtype=torch.int32).pin_memory()
                _lib.check(L.odam_assoc_hungarian(_lib.ptr(Z), ctypes.c_int(n_tracks), ctypes.c_int(n_det), ctypes.c_int(n_det + 1),
                                                  ctypes.c_double(float(threshold)), ctypes.c_int(1), _lib.ptr(self._pin_match),
                                                  ctypes.c_void_p(self._pin_match.data_ptr() + 4 * 31),
                                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "odam_assoc_hungarian")
                self._match = (self._pin_match, n_det)
        if _pinned or self._match is not None:
            torch.cuda.current_stream(dev).synchronize()
        return Z

    def assignment_checked(self, tracks, detections, n_tracks, n_det, threshold=None):
        """assignment() + synchronisation + the lost-launch check: if the persistent matching kernel could not get all its
        workgroups resident within the barrier bound (a foreign kernel held the CUs), this frame is run again through the
        launch sequence in the same process and counted in self.lost_launches.  The reference cannot fail this way
        (associator.py:202-268), so neither may the drop-in.  Returns Z on the host when the detections came from the host (the
        kernels then read them from, and write Z to, pinned memory: no copy commands), on the device otherwise."""
        self._pin_mode = True
        try:
            Z = self.assignment(tracks, detections, n_tracks, n_det, threshold=threshold)
            torch.cuda.current_stream(self.device).synchronize()
            n = ctypes.c_uint(0)
            _lib.check(_lib.lib().odam_assoc_lost_launches(self._handle(), ctypes.byref(n)), "odam_assoc_lost_launches")
            if n.value != self._lost_seen:
                self._lost_seen = n.value
                self.lost_launches += 1
                self._lost_in_a_row = getattr(self, "_lost_in_a_row", 0) + 1
                Z = self.assignment(tracks, detections, n_tracks, n_det, sequence=True, threshold=threshold)
                torch.cuda.current_stream(self.device).synchronize()
                if self._lost_in_a_row >= 3:
                    # three abandoned launches in a row are not a busy device but a condition that will not pass (the row-partitioned
                    # kernel's placement check, a stream confined to a share of the compute units): stay on the launch sequence
                    self.set_persistent(False)
                    self.persistent_given_up = True
            else:
                self._lost_in_a_row = 0
        finally:
            self._pin_mode = False
        return Z if Z.is_cuda else Z.clone()      # (the pinned buffer is the next frame's)

    # ---- one frame's device work behind one native call (include/odam_assoc.h, odam_assoc_step) -------------------------------------
    def step_launch(self, win, fitter, app_ids, app_rows14, n_tracks, cam21, cam_azi, img_w, img_h, det32T):
        """Enqueue, for one frame: the mirror append of what the previous frame attached (app_ids / app_rows14, may be empty), the
        track input for this frame's camera and the association forward on it -- OdamProcess's fast path (processor.py::process_frames).
        det32T [79, n_det] float32 (the frame's detection rows transposed, as process_frame hands them to the collater).  Nothing is
        synchronised here; step_finish() waits and returns the log assignment."""
        dev = self.device
        n_det = det32T.shape[1]
        if n_tracks > self.max_tracks:
            self._grow(n_tracks)
        if getattr(self, "_pin", None) is None:
            self._pin = (torch.empty(79 * 30, dtype=torch.float32).pin_memory(), torch.empty((self.MAX_TRACKS + 1) * 31, dtype=torch.float32).pin_memory())
        de = self._pin[0].view(79, 30).numpy()
        de[:, n_det:] = -1.0                      # the collater's padding (processor.py:49-68)
        de[:, :n_det] = det32T
        buf = getattr(self, "_tracks_buf", None)
        if buf is None or buf.shape[0] < n_tracks:
            self._tracks_buf = buf = torch.empty(max(64, 2 * n_tracks), 79, TrackWindows.WINDOW, device=dev, dtype=torch.float32)
        L = _lib.lib()
        n_app = len(app_ids)
        ids = np.ascontiguousarray(app_ids, np.int32)
        rows = np.ascontiguousarray(app_rows14, np.float64)
        # (per-frame host cost matters here -- the device is idle until the call below: the device context is entered only when another
        # device is current, and the completion event is one object, recorded again every frame)
        dev_t = torch.device(dev)
        ctx = torch.cuda.device(dev_t) if torch.cuda.current_device() != (dev_t.index or 0) else contextlib.nullcontext()
        with ctx, fitter._lock:
            cur = torch.cuda.current_stream(dev_t)
            stream = cur.cuda_stream
            if fitter._last is not None and fitter._last[0] != stream:
                cur.wait_event(fitter._last[1])
            _lib.check(L.odam_assoc_step(self._handle(), win._h, fitter._h, ctypes.c_int(n_app), ctypes.c_void_p(ids.ctypes.data),
                                         ctypes.c_void_p(rows.ctypes.data), ctypes.c_int(n_tracks), ctypes.c_void_p(cam21.ctypes.data),
                                         ctypes.c_double(cam_azi), ctypes.c_double(img_w), ctypes.c_double(img_h), _lib.ptr(self._pin[0]),
                                         ctypes.c_int(n_det), _lib.ptr(buf), _lib.ptr(self._pin[1]), ctypes.c_void_p(stream)), "odam_assoc_step")
            ev = getattr(self, "_step_ev", None)
            if ev is None:
                ev = self._step_ev = torch.cuda.Event()
            ev.record(cur)
            fitter._last = (stream, ev)
        self._step = (n_tracks, n_det, fitter)

    def step_finish(self):
        """Wait for the frame enqueued by step_launch and return its log assignment [(T + 1), (n_det + 1)] -- a view of the pinned buffer,
        valid until the next step_launch.  An abandoned persistent launch is re-run through the launch sequence, as in assignment_checked."""
        n_tracks, n_det, fitter = self._step
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()
        fitter._last = None                    # synchronised: no cross-stream wait is owed
        n = ctypes.c_uint(0)
        _lib.check(_lib.lib().odam_assoc_lost_launches(self._handle(), ctypes.byref(n)), "odam_assoc_lost_launches")
        if n.value != self._lost_seen:
            self._lost_seen = n.value
            self.lost_launches += 1
            self._lost_in_a_row = getattr(self, "_lost_in_a_row", 0) + 1
            self._pin_mode = True
            try:
                self.assignment(self._tracks_buf[:n_tracks], self._pin[0].view(79, 30), n_tracks, n_det, sequence=True)
                cur.synchronize()
            finally:
                self._pin_mode = False
            if self._lost_in_a_row >= 3:
                self.set_persistent(False)
                self.persistent_given_up = True
        else:
            self._lost_in_a_row = 0
        return self._pin[1][:(n_tracks + 1) * (n_det + 1)].view(n_tracks + 1, n_det + 1)

    def __call__(self, in_data, threshold, eval_only=False, device="cuda"):
        """Associator.forward for the batch-of-one call of OdamProcess (associator.py:202-268, eval_only=True)"""
        valid = in_data["valid_list"]
        if len(valid) != 1:
            raise _lib.OdamError("batch size 1 only (what OdamProcess.process_frame passes)")
        n_tracks, n_det = valid[0]
        # ONE download of the log assignment: both consumers -- the Hungarian step here and process_frame's score matrix
        # (processor.py:337-339: predictions['pred'][0][0].cpu().exp()) -- read it on the host
        self._match = None      # (set by assignment() when the Hungarian step ran on the device behind it)
        Z = self.assignment_checked(in_data["tracks"], in_data["detections"][0], n_tracks, n_det, threshold=threshold).cpu()
        pred = [Z[None]]
        m = getattr(self, "_match", None)
        if m is not None and int(m[0][31]) == 0:      # solved on the device behind the Sinkhorn loop (status word 0)
            match = [m[0][:n_det].numpy().astype(np.float64)]
        else:
            match = [hungarian_matching(Z[:-1, :-1].exp(), threshold)]
        return {"pred": pred, "loss": 0, "matches": match}

    forward = __call__


def build(args):
    """src/models/associator.py:330-340"""
    g = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
    try:
        max_tracks = int(g("max_tracks"))      # optional: initial workspace size (grows on demand)
    except (KeyError, AttributeError):
        max_tracks = 256
    return Associator({"GNN_layers": g("GNN_layers"), "self_GNN_layers": g("self_GNN_layers"),
                       "sinkhorn_iterations": g("sinkhorn_iterations")}, max_tracks=max_tracks)
