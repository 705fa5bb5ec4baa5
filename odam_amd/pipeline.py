"""One scene through the whole chain on N ranks -- what the reference's driver does on one process
(likojack/ODAM src/scripts/run_processor.py:70-83):

    for frame: process_frame(rgb, frame_id, T_wc)        # detect -> associate -> attach   (src/processor.py:306-345)
    out = optim_process(tracks); tracks = merge_process(out); out = optim_process(tracks)

with the two data-parallel halves sharded and everything in between replayed identically on every rank:

    chunk of frames -> each rank detects ITS frames of the chunk -> all-gather of the detection blocks (collective #1)
                    -> every rank replays process_frame over the chunk's frames in order (association is sequential)
    tracks (identical on every rank) -> object_shard by observation count -> each rank fits ITS objects
                    -> all-gather of parameters + boxes (collective #2) -> every rank holds the full result dict
    merge_process on every rank (deterministic host code) -> second sharded fit pass.

`run_scene` is the product entry; the ranks are the processes of the current torch.distributed group (one per GPU, backend
"nccl" = RCCL; "gloo" in the tests), or a world of one without a group.  The result dict has the reference's pickle layout
(run_processor.py:85-92) on EVERY rank.  Detection rows travel as float32 -- they are float32 values widened to float64 by
run_detector (processor.py:276-288), so nothing is lost -- and float64 boxes travel bit for bit (as pairs of 32-bit words): an
N-rank run returns what the one-rank run returns, bit for bit, when the detector's kernel choice does not depend on how many
frames a rank holds (odam_config key cg.pin = 1; otherwise to the fp32 reorder noise of different tile shapes, ~1e-5).
"""
import numpy as np

from . import multi_view, parallel

# one fitted object on the wire: 9 parameters | fitted flag | class | bboxes_qc 8x3 float64 | bboxes_dl 8x3 float64 (as float32 words)
_FIT_COLS = 9 + 1 + 1 + 48 + 48


def _f64_words(a):
    """float64 [n, 24] -> float32 [n, 48] holding the same bytes (a gather moves bytes: the values arrive bit for bit)"""
    return np.ascontiguousarray(a, np.float64).reshape(len(a), 24).view(np.float32)


def detect_gathered(detect, first, end, device=None, force=False):
    """Frames first .. end-1 of the scene: this rank detects its contiguous share (parallel.frame_shard of the span), the
    blocks are all-gathered, every rank returns the rows of ALL frames of the span in frame order (list of [n, 79] float64).
    `detect(f0, f1)` -> list (per frame) of [n, 79] rows for scene frames f0 .. f1-1."""
    rank, ws = parallel.world()
    n = end - first
    s, e = parallel.frame_shard(n, rank, ws)
    rows = detect(first + s, first + e) if e > s else []
    blk, cnt = parallel.pack_detections(rows)
    blk, cnt = parallel.allgather_detections(blk, cnt, n, device, force=force)
    return parallel.unpack_detections(blk, cnt)


def sharded_optim(proc, tracks, device=None, force=False):
    """optim_process (src/scripts/run_multi_view.py:22-76) with the objects dealt over the ranks: each rank fits its share
    (longest tracks first, round-robin), one all-gather brings parameters and boxes of every object to every rank.
    Returns the reference's dict {"tracks", "bboxes_qc", "bboxes_dl", "quadrics"} for ALL objects."""
    rank, ws = parallel.world()
    n = len(tracks)
    owners = [parallel.object_shard([len(t) for t in tracks], r, ws) for r in range(ws)]
    mine = owners[rank]
    local = np.zeros((len(mine), _FIT_COLS), np.float32)
    if mine:
        out = proc.optim_process_params([tracks[i] for i in mine])
        local[:, :9] = out["params"]
        local[:, 9] = out["fitted"]
        local[:, 10] = [q.obj_class for q in out["quadrics"]]
        local[:, 11:59] = _f64_words(np.asarray(out["bboxes_qc"], np.float64))
        local[:, 59:107] = _f64_words(np.asarray(out["bboxes_dl"], np.float64))
        mine_quadrics = dict(zip(mine, out["quadrics"]))
    else:
        mine_quadrics = {}
    rows = parallel.allgather_rows(local, n, owners, device, force=force)
    fitter = proc._fitter()
    qc = np.ascontiguousarray(rows[:, 11:59]).view(np.float64).reshape(n, 8, 3)
    dl = np.ascontiguousarray(rows[:, 59:107]).view(np.float64).reshape(n, 8, 3)
    quadrics = []
    for i in range(n):
        # the objects this rank fitted keep their sampled surface; the others re-sample it from the parameters on demand
        # (compute_ellipsoid_points: the same sampler and surface code on the same nine floats)
        quadrics.append(mine_quadrics.get(i) or multi_view.SuperQuadric(rows[i, :9], int(rows[i, 10]), None, fitter))
    return {"tracks": tracks, "bboxes_qc": [qc[i] for i in range(n)], "bboxes_dl": [dl[i] for i in range(n)],
            "quadrics": quadrics, "params": rows[:, :9].copy(), "fitted": rows[:, 9] != 0}


def _associate(proc, frame_ids, T_wcs, c0, rows):
    """the sequential association loop (src/processor.py:306-345) over the frames c0 ... of a chunk whose detections are at hand: as one
    process_frames call where the process object has it (the host's work then sits under the device's, processor.py), frame by frame otherwise"""
    if hasattr(proc, "process_frames"):
        n = len(rows)
        proc.process_frames([int(frame_ids[c0 + i]) for i in range(n)], [T_wcs[c0 + i] for i in range(n)], rows)
    else:
        for i, r in enumerate(rows):
            proc.process_frame(None, int(frame_ids[c0 + i]), T_wcs[c0 + i], detections=r)


def run_scene(proc, n_frames, frame_ids, T_wcs, detect=None, frames=None, chunk=0, device=None, force=False, stages=None, overlap=None):
    """The driver loop of run_processor.py:70-83 for one scene on the ranks of the current process group.

    proc      OdamProcess after init_sequence (its detector is used unless `detect` is given; its associator and fitter always)
    detect    optional callable (f0, f1) -> per-frame [n, 79] rows for scene frames f0 .. f1-1 (e.g. a batched, device-resident
              path); default: proc.detect_frames over frames[f0:f1]
    chunk     frames per all-gather (0 = the whole scene in one): association of a chunk starts when its detections arrive
    stages    optional dict that receives wall-clock seconds per stage (detect + gather, associate, fit 1, merge, fit 2)
    overlap   detection + all-gather of chunk i + 1 run on a worker thread while this thread replays process_frame over chunk i
              ("chunking lets association overlap detection"; src/processor.py:306-345 is the loop that is replicated).
              None = on in a world of more than one rank with more than one chunk: there a rank's share of a chunk's detection is
              1 / N of it while the replicated association is not, so the chain runs at max(associate, detect / N) instead of their
              sum; on ONE GPU the association's short launches wait behind the detector's queued ones (profiles/
              r4_experiments_not_kept.txt #12, #13) and the overlap returns nothing.  Only the worker issues collectives while it
              runs, in chunk order on every rank; results are identical with and without (the same calls on the same data).
    Returns the second optim_process dict (what the reference pickles), identical on every rank; proc.tracks holds the
    unmerged tracks."""
    import sys
    import time
    if detect is None:
        if frames is None:
            raise ValueError("run_scene: give `frames` or a `detect` callable")

        def detect(f0, f1):
            return [np.asarray(r, np.float64).reshape(-1, 79) for r in proc.detect_frames(list(frames[f0:f1]), list(frame_ids[f0:f1]))]
    t = {"detect_gather": 0.0, "associate": 0.0}
    spans = parallel.chunk_spans(n_frames, chunk or n_frames)
    if overlap is None:
        overlap = parallel.world()[1] > 1 and len(spans) > 1
    if overlap and len(spans) > 1:
        import queue
        import threading
        import torch
        q = queue.Queue(maxsize=2)      # at most two chunks of detections ahead of the association loop
        cancel = threading.Event()
        # HIP's current device is per THREAD and a new thread starts on device 0: resolve the device here, on the caller's thread, to a
        # full name ('cuda:N') and make it the worker's current device -- with device=None, rank r's worker would otherwise resolve
        # 'cuda' (and parallel._dev_key(None)) to GPU 0 and put its collective tensors there
        if device is None:
            wdev = parallel._dev_key(None) if (torch.cuda.is_available() and torch.cuda.is_initialized()) else None
        else:
            wdev = str(device) if str(device) == "cpu" else parallel._dev_key(device)

        def put(item):
            while not cancel.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                if wdev is not None and wdev != "cpu":
                    torch.cuda.set_device(torch.device(wdev))
                for c0, c1 in spans:
                    if cancel.is_set():      # the consumer failed: issue no further collective from this rank
                        return
                    s0 = time.perf_counter()
                    rows = detect_gathered(detect, c0, c1, device, force)
                    if not put((c0, rows, time.perf_counter() - s0)):
                        return
                put(None)
            except BaseException as e:      # noqa: BLE001 -- handed to the consumer, which re-raises it
                put(e)

        th = threading.Thread(target=produce, name="odam-detect", daemon=True)
        th.start()
        waited = 0.0
        try:
            while True:
                s0 = time.perf_counter()
                item = q.get()
                waited += time.perf_counter() - s0
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                c0, rows, dt_detect = item
                s1 = time.perf_counter()
                _associate(proc, frame_ids, T_wcs, c0, rows)
                t["associate"] += time.perf_counter() - s1
                t["detect_gather"] += dt_detect
        finally:
            # on an exception (here or in the producer) the worker must not keep issuing collectives on this rank alone, nor stay blocked on
            # a full queue: tell it to stop, make room, wait for it.  (A collective it is INSIDE completes only if the peers reach it too;
            # the join is bounded so that this rank's exception surfaces either way.)
            cancel.set()
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=None if sys.exc_info()[0] is None else 30.0)
        t["overlapped"] = True
        t["associate_waited_for_detections"] = waited      # what the association loop did NOT hide
    else:
        for c0, c1 in spans:
            s0 = time.perf_counter()
            rows = detect_gathered(detect, c0, c1, device, force)
            s1 = time.perf_counter()
            _associate(proc, frame_ids, T_wcs, c0, rows)
            t["detect_gather"] += s1 - s0
            t["associate"] += time.perf_counter() - s1
    s0 = time.perf_counter()
    out = sharded_optim(proc, proc.tracks, device, force)
    s1 = time.perf_counter()
    merged = proc.merge_process(out)
    s2 = time.perf_counter()
    out2 = sharded_optim(proc, merged, device, force)
    t.update(fit1=s1 - s0, merge=s2 - s1, fit2=time.perf_counter() - s2)
    if stages is not None:
        stages.update(t)
        stages["first_pass"] = out
    return out2
