"""Data-parallel layout of a scene over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The path shards twice: detection is stateless per frame (src/processor.py:259-289),
super-quadric fits are independent per object (src/scripts/run_multi_view.py:44-69); association between
them is sequential and replicated.  Two tiny collectives carry everything:
  1. all-gather of fixed-shape detection blocks  float32[F_local, 30, 15] + int32[F_local]   (before association)
  2. all-gather of fitted parameters / boxes     float32[n_local, 9] (+ [n_local, 8, 3])       (after the fit)
This replaces the reference's pickle-based variable-size gather (src/utils/misc.py:90-130), which its
inference path never calls.  Messages are <= a few MB per scene: latency-bound, one collective each.
"""
import numpy as np
import torch
import torch.distributed as dist

MAX_DETS = 30   # processor.py:318-319
DET_COLS = 15   # the 64 shape-code columns are constant -1 with no_code=True (processor.py:286-287)


_NATIVE = {}      # device string -> (handle, world, rank): the library's own RCCL communicator (include/odam_comm.h)


def use_native_collectives(device):
    """Route both all-gathers through the library's C ABI (include/odam_comm.h: ncclAllGather on the librccl the process has
    mapped) instead of torch.distributed.  Collective over the ranks of the current process group: rank 0 draws the RCCL unique
    id, torch.distributed carries its 128 bytes to the others (any backend), every rank creates its communicator on `device`.
    Needs an initialised process group (a world of one included).  Returns the communicator's (world, rank)."""
    import ctypes
    from . import _lib
    device = _dev_key(device)
    if device in _NATIVE:
        return _NATIVE[device][1:]
    rank, ws = dist.get_rank(), dist.get_world_size()
    L = _lib.lib()
    box = [None]
    if rank == 0:
        buf = (ctypes.c_ubyte * 128)()
        _lib.check(L.odam_comm_unique_id(buf), "odam_comm_unique_id")
        box[0] = bytes(buf)
    dist.broadcast_object_list(box, src=0)
    h = ctypes.c_void_p()
    with torch.cuda.device(torch.device(device)):
        _lib.check(L.odam_comm_create((ctypes.c_ubyte * 128).from_buffer_copy(box[0]), ctypes.c_int(ws), ctypes.c_int(rank),
                                      ctypes.byref(h)), "odam_comm_create")
    _NATIVE[device] = (h, ws, rank)
    return ws, rank


def close_native_collectives():
    from . import _lib
    for h, _, _ in _NATIVE.values():
        _lib.lib().odam_comm_destroy(h)
    _NATIVE.clear()


def _dev_key(device):
    """'cuda' / None -> the current device's full name ('cuda:0'); a torch.device or string otherwise"""
    d = torch.device("cuda" if device is None else device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def _native(device):
    """The library's communicator for `device`, or None while native collectives are off.  Once use_native_collectives() has run,
    the mode is process-wide: every collective of this module goes through the library's communicator, and a device without one is
    an error -- never a silent fall-back to torch.distributed, which would leave this rank in a different collective than its
    peers (a deadlock) and, under a gloo group, hand CUDA tensors to gloo."""
    if not _NATIVE:
        return None
    d = None if device is None else torch.device(device)
    if (d is None or d.type == "cpu" or d.index is None) and len(_NATIVE) == 1:
        # no device named (or host staging asked for): the communicator's OWN device, not the calling thread's current one -- a helper
        # thread starts on device 0, and a rank-dependent error here would strand the peers inside the collective
        key = next(iter(_NATIVE))
        return _NATIVE[key] + (key,)
    key = _dev_key(device)
    if key not in _NATIVE:
        from . import _lib
        raise _lib.OdamError(f"native collectives are on (use_native_collectives) but no communicator exists for {key}; "
                             f"have {sorted(_NATIVE)}")
    return _NATIVE[key] + (key,)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def frame_shard(n_frames, rank, world_size):
    """contiguous block of frame indices owned by `rank` (blocks differ by at most one frame)"""
    base, rem = divmod(n_frames, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def chunk_spans(n_frames, chunk):
    """[(first, end)] of consecutive chunks of `chunk` frames (the last one ragged).  Gathering detections chunk by chunk --
    each chunk dealt over the ranks with frame_shard(end - first, rank, world) and gathered with allgather_detections --
    hands the sequential, replicated association loop (src/processor.py:306-345) the frames of chunk i while the detectors
    work on chunk i + 1, instead of after the whole scene."""
    return [(c0, min(n_frames, c0 + chunk)) for c0 in range(0, n_frames, chunk)]


def pack_detections(per_frame_rows):
    """list (per local frame) of lists of 79-float rows -> (float32[F,30,15], int32[F]) truncated to 30 like
    process_frame does"""
    F = len(per_frame_rows)
    blk = np.full((F, MAX_DETS, DET_COLS), -1.0, np.float32)
    cnt = np.zeros(F, np.int32)
    for f, rows in enumerate(per_frame_rows):     # rows: list of 79-float lists, or an [n, 79] array
        n = min(len(rows), MAX_DETS)
        cnt[f] = n
        if n:
            blk[f, :n] = (rows if isinstance(rows, np.ndarray) else np.asarray(rows, np.float64))[:n, :DET_COLS]
    return blk, cnt


def unpack_detections(blk, cnt):
    """inverse of pack_detections: list of [n,79] float64 arrays (code columns restored to -1)"""
    out = []
    for f in range(len(cnt)):
        n = int(cnt[f])
        rows = np.full((n, 79), -1.0)
        rows[:, :DET_COLS] = blk[f, :n].astype(np.float64)
        out.append(rows)
    return out


def allgather_detections(blk, cnt, n_frames, device=None, force=False):
    """Collective #1.  Every rank passes its block for frame_shard(n_frames, rank, world); returns the
    blocks of the whole scene in frame order on every rank.  One message per rank: the block and its per-frame counts
    travel in ONE float32 tensor [F_pad, 30 * 15 + 1] (the count in the last column; exact, it is <= 30) -- one upload,
    one all_gather_into_tensor, one download.  `force`: run the collective even in a world of one (exercises the RCCL
    call path on a single GPU)."""
    rank, ws = world()
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return blk, cnt
    dev = device or ("cuda" if (dist.get_backend() == "nccl" or _NATIVE) else "cpu")
    per = -(-n_frames // ws)  # padded shard length so all_gather_into_tensor sees equal shapes
    W = MAX_DETS * DET_COLS
    nat = _native(dev)
    if nat is not None:
        dev = nat[3]
    if nat is not None:      # the library's own collective (include/odam_comm.h, odam_allgather_detections): block and counts stay separate
        import ctypes
        from . import _lib
        hb = np.full((per, MAX_DETS, DET_COLS), -1.0, np.float32)
        hc = np.zeros(per, np.int32)
        hb[:len(cnt)] = blk; hc[:len(cnt)] = cnt
        db, dc = torch.from_numpy(hb).to(dev), torch.from_numpy(hc).to(dev)
        ab = torch.empty((ws, per, MAX_DETS, DET_COLS), dtype=torch.float32, device=dev)
        ac = torch.empty((ws, per), dtype=torch.int32, device=dev)
        with torch.cuda.device(torch.device(dev)):
            _lib.check(_lib.lib().odam_allgather_detections(nat[0], _lib.ptr(db), _lib.ptr(dc), ctypes.c_int(per), _lib.ptr(ab), _lib.ptr(ac),
                                                            ctypes.c_void_p(torch.cuda.current_stream(torch.device(dev)).cuda_stream)),
                       "odam_allgather_detections")
        gb, gc = ab.cpu().numpy(), ac.cpu().numpy()
        ob, oc = [], []
        for r in range(ws):
            s, e = frame_shard(n_frames, r, ws)
            ob.append(gb[r, :e - s]); oc.append(gc[r, :e - s])
        return np.concatenate(ob), np.concatenate(oc)
    host = np.full((per, W + 1), -1.0, np.float32)
    host[:, W] = 0.0
    host[:len(cnt), :W] = blk.reshape(len(cnt), W)
    host[:len(cnt), W] = cnt
    mine = torch.from_numpy(host).to(dev)
    allr = torch.empty((ws * per, W + 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(allr, mine)
    g = allr.cpu().numpy()
    ob, oc = [], []
    for r in range(ws):
        s, e = frame_shard(n_frames, r, ws)
        ob.append(g[r * per: r * per + (e - s), :W].reshape(e - s, MAX_DETS, DET_COLS))
        oc.append(g[r * per: r * per + (e - s), W].astype(np.int32))
    return np.concatenate(ob), np.concatenate(oc)


def object_shard(view_counts, rank, world_size):
    """objects sorted by view count (descending) dealt round-robin: the indices owned by `rank`"""
    order = np.argsort(-np.asarray(view_counts), kind="stable")
    return [int(i) for i in order[rank::world_size]]


def allgather_rows(local, n_total, owner_lists, device=None, force=False):
    """Collective #2.  `local` float32[n_local, ...] rows for owner_lists[rank] (same trailing shape on all
    ranks) -> float32[n_total, ...] in object order on every rank."""
    rank, ws = world()
    local = np.ascontiguousarray(local, np.float32)
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        out = np.zeros((n_total,) + local.shape[1:], np.float32)
        out[owner_lists[0]] = local
        return out
    dev = device or ("cuda" if (dist.get_backend() == "nccl" or _NATIVE) else "cpu")
    per = max(len(o) for o in owner_lists)
    nat = _native(dev)
    if nat is not None:
        dev = nat[3]
        import ctypes
        from . import _lib
        pad = torch.zeros((per,) + local.shape[1:], dtype=torch.float32, device=dev)
        if len(local):
            pad[:len(local)] = torch.from_numpy(local).to(dev)
        g = torch.empty((ws * per,) + local.shape[1:], dtype=torch.float32, device=dev)
        with torch.cuda.device(torch.device(dev)):
            _lib.check(_lib.lib().odam_allgather_f32(nat[0], _lib.ptr(pad), ctypes.c_longlong(pad.numel()), _lib.ptr(g),
                                                     ctypes.c_void_p(torch.cuda.current_stream(torch.device(dev)).cuda_stream)), "odam_allgather_f32")
        g = g.cpu().numpy()
        out = np.zeros((n_total,) + local.shape[1:], np.float32)
        for r in range(ws):
            idx = owner_lists[r]
            out[idx] = g[r * per: r * per + len(idx)]
        return out
    pad = torch.zeros((per,) + local.shape[1:], dtype=torch.float32, device=dev)
    if len(local):
        pad[:len(local)] = torch.from_numpy(local).to(dev)
    g = torch.empty((ws * per,) + local.shape[1:], dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(g, pad)
    g = g.cpu().numpy()
    out = np.zeros((n_total,) + local.shape[1:], np.float32)
    for r in range(ws):
        idx = owner_lists[r]
        out[idx] = g[r * per: r * per + len(idx)]
    return out
