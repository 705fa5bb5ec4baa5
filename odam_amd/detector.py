"""Host side of the detector: a drop-in for what OdamProcess needs from the reference's DETR object
(likojack/ODAM src/models/detr.py): `detector(img_tensor) -> dict`, `detector.postprocess(...)`,
`load_state_dict` with the reference's key names, `.to(device)`, `.eval()`, and `build(cfg)`.

The arithmetic runs in the gfx950 library behind include/odam_detr.h; this file keeps only what the
reference keeps on the host (thresholding, greedy nms_3d, list-of-numpy packaging, detr.py:124-159,
161-205) plus the constant sine position embedding (position_encoding.py:26-46)."""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .weights import RESNET_BLOCKS


class _Cfg(ctypes.Structure):
    _fields_ = [("resnet_blocks", ctypes.c_int * 4), ("hidden_dim", ctypes.c_int), ("nheads", ctypes.c_int),
                ("dim_feedforward", ctypes.c_int), ("enc_layers", ctypes.c_int), ("dec_layers", ctypes.c_int),
                ("num_queries", ctypes.c_int), ("num_classes1", ctypes.c_int), ("angle_bins", ctypes.c_int),
                ("max_batch", ctypes.c_int), ("img_h", ctypes.c_int), ("img_w", ctypes.c_int), ("dtype", ctypes.c_int),
                ("pre_norm", ctypes.c_int), ("dilation", ctypes.c_int)]


def _axis_phase_table(counts, channels, temperature):
    """[..., channels] float32: normalised running token count along one image axis times 2 pi, divided by the channel's
    wavelength; even channels carry the sine and odd channels the cosine of the SAME wavelength (pairs share an exponent).
    `counts` float32 [...], already divided by the axis total.  Every operation is the float32 torch CPU op the reference
    applies at this point (position_encoding.py:31-43), so the table can be compared bit for bit."""
    pair = torch.arange(channels, dtype=torch.float32).div(2, rounding_mode="floor")
    wavelength = torch.pow(float(temperature), pair * 2 / channels)
    phase = counts.unsqueeze(-1) / wavelength
    table = torch.empty_like(phase)
    table[..., 0::2] = torch.sin(phase[..., 0::2])
    table[..., 1::2] = torch.cos(phase[..., 1::2])
    return table


def sine_position_embedding(h, w, num_pos_feats=128, temperature=10000, mask=None):
    """Sine position table of an h x w token grid -> [h*w, 2*num_pos_feats] float32, row-major tokens, the vertical half of
    the channels first (what PositionEmbeddingSine(128, normalize=True) returns, position_encoding.py:26-46; checked bit for
    bit against the reference-run table tests/golden/detr_small.npz["pos_25x34"]).  `mask` [h,w] bool (True = padding)
    for an image padded inside a mixed-size batch; None = no padding.

    A token's coordinate is the count of unpadded tokens up to and including it along the axis, over the count of the
    whole row / column (+1e-6), times 2 pi.  Without padding the two coordinates depend on the row resp. the column only, so
    two 1-D tables are built and broadcast; with padding the counts are 2-D."""
    two_pi = 2 * math.pi
    out = torch.empty(h, w, 2 * num_pos_feats, dtype=torch.float32)
    if mask is None:
        rows = torch.arange(1, h + 1, dtype=torch.float32)
        cols = torch.arange(1, w + 1, dtype=torch.float32)
        out[:, :, :num_pos_feats] = _axis_phase_table(rows / (rows[-1] + 1e-6) * two_pi, num_pos_feats, temperature)[:, None, :]
        out[:, :, num_pos_feats:] = _axis_phase_table(cols / (cols[-1] + 1e-6) * two_pi, num_pos_feats, temperature)[None, :, :]
    else:
        live = (~mask.reshape(h, w)).to(torch.float32)
        down, across = live.cumsum(0), live.cumsum(1)
        out[:, :, :num_pos_feats] = _axis_phase_table(down / (down[-1:, :] + 1e-6) * two_pi, num_pos_feats, temperature)
        out[:, :, num_pos_feats:] = _axis_phase_table(across / (across[:, -1:] + 1e-6) * two_pi, num_pos_feats, temperature)
    return out.reshape(h * w, 2 * num_pos_feats)


def learned_position_embedding(row_embed, col_embed, h, w):
    """PositionEmbeddingLearned for an h x w grid -> [h*w, 256] float32 (position_encoding.py:49-74): token (y, x) carries
    cat(col_embed[x], row_embed[y]); the tables hold 50 rows (h, w <= 50), and the padding mask plays no part."""
    if h > row_embed.shape[0] or w > col_embed.shape[0]:
        raise _lib.OdamError("learned position embedding: the token grid exceeds the embedding tables (50 x 50)")
    x_emb, y_emb = col_embed[:w], row_embed[:h]
    pos = torch.cat([x_emb.unsqueeze(0).repeat(h, 1, 1), y_emb.unsqueeze(1).repeat(1, w, 1)], dim=-1)
    return pos.reshape(h * w, -1).to(torch.float32).contiguous()


class Detector:
    """DETR forward on the GPU library.  One native handle per input size (the workspace and the
    position table depend on it), created lazily at the first call with that size."""

    def __init__(self, backbone="resnet50", hidden_dim=256, nheads=8, dim_feedforward=2048, enc_layers=6,
                 dec_layers=6, num_queries=100, num_classes=18, angle_bins=30, max_batch=8, device="cuda:0",
                 n_streams=2, dtype="fp32", pre_norm=False, position_embedding="sine", dilation=False):
        self.n_streams = max(1, int(n_streams))
        if position_embedding not in ("sine", "v2", "learned", "v3"):      # build_position_encoding (position_encoding.py:77-87)
            raise ValueError(f"not supported {position_embedding}")
        self.learned_pos = position_embedding in ("learned", "v3")
        self.pre_norm = bool(pre_norm)
        self.dilation = bool(dilation)
        self.resize = (800, 1333)   # get_transforms(): shortest side, longest-side cap (transforms.py:281-290)
        if dtype not in ("fp32", "bf16"):
            raise ValueError("dtype must be 'fp32' (parity mode) or 'bf16' (BASELINE config 4)")
        self.dtype = dtype
        self.arch = dict(backbone=backbone, hidden_dim=hidden_dim, nheads=nheads, dim_feedforward=dim_feedforward,
                         enc_layers=enc_layers, dec_layers=dec_layers, num_queries=num_queries,
                         num_classes1=num_classes + 1, angle_bins=angle_bins)
        self.num_queries = num_queries
        self.max_batch = max_batch
        self.device = torch.device(device)
        self._sd = None
        self._handles = {}

    # ---- nn.Module-like surface used by run_processor.py:30-34 -------------------------------------
    def to(self, device):
        new = torch.device("cuda:0" if str(device) == "cuda" else device)
        if new != self.device:
            self.close()          # handles, streams and scratch belong to the old device
            for name in ("_streams", "_mscratch", "_xform", "_stage", "_copy_stream", "_pinned"):
                if hasattr(self, name):
                    setattr(self, name, None)
        self.device = new
        return self

    def cuda(self):
        return self.to("cuda:0")

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in state_dict.items()
                    if torch.is_tensor(v) and not k.endswith("num_batches_tracked")}
        self.close()
        return self

    def close(self):
        for h in self._handles.values():
            _lib.lib().odam_detr_destroy(h)
        self._handles = {}
        self._streams = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self, H, W, idx=0):
        key = (H, W, idx)
        if key in self._handles:
            return self._handles[key]
        if self._sd is None:
            raise _lib.OdamError("Detector: load_state_dict() must be called before the first forward")
        L = _lib.lib()
        a = self.arch
        cfg = _Cfg()
        for i, n in enumerate(RESNET_BLOCKS[a["backbone"]]):
            cfg.resnet_blocks[i] = n
        for k in ("hidden_dim", "nheads", "dim_feedforward", "enc_layers", "dec_layers", "num_queries", "num_classes1",
                  "angle_bins"):
            setattr(cfg, k, a[k])
        cfg.max_batch, cfg.img_h, cfg.img_w = self.max_batch, H, W
        cfg.dtype = 1 if self.dtype == "bf16" else 0
        cfg.pre_norm = 1 if self.pre_norm else 0
        cfg.dilation = 1 if self.dilation else 0
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.odam_detr_create(ctypes.byref(cfg), ctypes.byref(h)), "odam_detr_create")
            try:
                self._fill_handle(h)
            except Exception:
                L.odam_detr_destroy(h)     # a half-built handle (e.g. a missing weight) is not kept
                raise
        self._handles[key] = h
        return h

    def _fill_handle(self, h):
        """weights (reference key names) + the position table of this input size, then finalize"""
        L = _lib.lib()
        a = self.arch
        fh, fw = ctypes.c_int(), ctypes.c_int()
        _lib.check(L.odam_detr_feature_hw(h, ctypes.byref(fh), ctypes.byref(fw)), "odam_detr_feature_hw")
        tensors = dict(self._sd)
        tensors["pos_embed"] = self._position_table(fh.value, fw.value)
        for name, t in tensors.items():
            shape = (ctypes.c_longlong * max(t.dim(), 1))(*t.shape)
            _lib.check(L.odam_detr_set_weight(h, name.encode(), ctypes.c_void_p(t.data_ptr()), shape, t.dim()),
                       f"odam_detr_set_weight({name})")
        _lib.check(L.odam_detr_finalize(h), "odam_detr_finalize")

    def _position_table(self, h, w, mask=None):
        if self.learned_pos:
            for k in ("backbone.1.row_embed.weight", "backbone.1.col_embed.weight"):
                if k not in self._sd:
                    raise _lib.OdamError(f"position_embedding='learned': the state dict has no {k}")
            return learned_position_embedding(self._sd["backbone.1.row_embed.weight"], self._sd["backbone.1.col_embed.weight"], h, w)
        return sine_position_embedding(h, w, self.arch["hidden_dim"] // 2, mask=mask)

    # ---- DETR.forward over images of different sizes (detr.py:64-65 -> misc.py:303-320) ----------------------
    def forward_nested(self, images):
        """list of [3,H_i,W_i] tensors -> the DETR output dict, as the reference computes it for a NestedTensor: images
        zero-padded at the top-left to the batch maximum, padding mask reduced to the token grid by nearest
        interpolation (backbone.py:79), per-image sine embedding from that mask, padded tokens excluded as attention
        keys in the encoder and in the decoder's cross-attention."""
        import torch.nn.functional as F
        imgs = [t.to(torch.float32) for t in images]
        B = len(imgs)
        if B > self.max_batch:
            raise _lib.OdamError("forward_nested: more images than max_batch")
        H, W = max(t.shape[1] for t in imgs), max(t.shape[2] for t in imgs)
        x = torch.zeros(B, 3, H, W, dtype=torch.float32)
        mask = torch.ones(B, H, W, dtype=torch.bool)
        for i, t in enumerate(imgs):
            x[i, :, :t.shape[1], :t.shape[2]] = t.cpu()
            mask[i, :t.shape[1], :t.shape[2]] = False
        h = self._handle(H, W)
        L = _lib.lib()
        fh, fw = ctypes.c_int(), ctypes.c_int()
        _lib.check(L.odam_detr_feature_hw(h, ctypes.byref(fh), ctypes.byref(fw)), "odam_detr_feature_hw")
        fh, fw = fh.value, fw.value
        m = F.interpolate(mask[None].float(), size=(fh, fw)).to(torch.bool)[0]            # backbone.py:79
        a = self.arch
        pos = torch.stack([self._position_table(fh, fw, mask=m[i]) for i in range(B)])
        dev = self.device
        d_x, d_m, d_pos = x.to(dev), m.reshape(B, fh * fw).to(torch.uint8).to(dev).contiguous(), pos.to(dev).contiguous()
        Q = a["num_queries"]
        mk = lambda n: torch.empty(B, Q, n, device=dev, dtype=torch.float32)
        out = {"pred_logits": mk(a["num_classes1"]), "pred_boxes": mk(4), "pred_angle": mk(a["angle_bins"]),
               "pred_offset": mk(2), "pred_size": mk(3), "pred_depth": mk(1), "pred_obj_features": mk(a["hidden_dim"])}
        with torch.cuda.device(dev):
            _lib.check(L.odam_detr_forward_masked(h, _lib.ptr(d_x), ctypes.c_int(B), _lib.ptr(d_m), _lib.ptr(d_pos),
                                                  _lib.ptr(out["pred_logits"]), _lib.ptr(out["pred_boxes"]),
                                                  _lib.ptr(out["pred_angle"]), _lib.ptr(out["pred_offset"]),
                                                  _lib.ptr(out["pred_size"]), _lib.ptr(out["pred_depth"]),
                                                  _lib.ptr(out["pred_obj_features"]),
                                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "odam_detr_forward_masked")
        out["_hw"] = (H, W)
        return out

    # ---- DETR.forward (detr.py:49-94) ---------------------------------------------------------------
    def __call__(self, samples):
        if isinstance(samples, (list, tuple)):
            if len({tuple(t.shape) for t in samples}) > 1:
                return self.forward_nested(samples)
            samples = torch.stack(list(samples))
        x = samples.to(self.device, torch.float32).contiguous()
        B, C, H, W = x.shape
        assert C == 3
        h = self._handle(H, W)
        a = self.arch
        Q = a["num_queries"]
        dev = self.device
        mk = lambda n: torch.empty(B, Q, n, device=dev, dtype=torch.float32)
        out = {"pred_logits": mk(a["num_classes1"]), "pred_boxes": mk(4), "pred_angle": mk(a["angle_bins"]),
               "pred_offset": mk(2), "pred_size": mk(3), "pred_depth": mk(1), "pred_obj_features": mk(a["hidden_dim"])}
        L = _lib.lib()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                sl = lambda t: _lib.ptr(t[b0:b1])
                _lib.check(L.odam_detr_forward(h, _lib.ptr(x[b0:b1]), ctypes.c_int(b1 - b0), sl(out["pred_logits"]),
                                               sl(out["pred_boxes"]), sl(out["pred_angle"]), sl(out["pred_offset"]),
                                               sl(out["pred_size"]), sl(out["pred_depth"]),
                                               sl(out["pred_obj_features"]), ctypes.c_void_p(stream)),
                           "odam_detr_forward")
        out["_hw"] = (H, W)
        return out

    forward = __call__

    # ---- transforms.py:281-290 on the device + DETR.forward ------------------------------------------
    def preprocess_u8(self, rgb, out=None, idx=0, stream=None):
        """uint8 [B,h,w,3] device frames -> float32 [B,3,H,W] (resize shortest side to `size`, /255, normalise):
        the reference's get_transforms() evaluated by the library, bit-identical to the PIL/torch host path."""
        from .transforms import MEAN, STD, target_size
        B, h, w, c = rgb.shape
        assert c == 3 and rgb.dtype == torch.uint8 and rgb.is_cuda and rgb.is_contiguous()
        H, W = target_size(w, h, *self.resize)
        if out is None:
            out = torch.empty(B, 3, H, W, device=self.device, dtype=torch.float32)
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        mean, std = (ctypes.c_float * 3)(*MEAN), (ctypes.c_float * 3)(*STD)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_detr_preprocess_u8(self._handle(H, W, idx), _lib.ptr(rgb), ctypes.c_int(B),
                                                          ctypes.c_int(h), ctypes.c_int(w), mean, std, _lib.ptr(out),
                                                          ctypes.c_void_p(st)), "odam_detr_preprocess_u8")
        return out

    def forward_u8(self, rgb):
        """raw uint8 [B,h,w,3] frames (device) -> the DETR output dict; B <= max_batch per preprocessing call"""
        outs = [self(self.preprocess_u8(rgb[b0:b0 + self.max_batch])) for b0 in range(0, rgb.shape[0], self.max_batch)]
        if len(outs) == 1:
            return outs[0]
        res = {k: torch.cat([o[k] for o in outs]) for k in outs[0] if k != "_hw"}
        res["_hw"] = outs[0]["_hw"]
        return res

    def debug_taps(self, B, H, W):
        """layer4 feature map [B,2048,h,w] and encoder memory [B,h*w,256] of the last forward (tests)."""
        hd = self._handle(H, W)
        L = _lib.lib()
        fh, fw = ctypes.c_int(), ctypes.c_int()
        L.odam_detr_feature_hw(hd, ctypes.byref(fh), ctypes.byref(fw))
        l4 = torch.empty(B, 2048, fh.value, fw.value, device=self.device)
        mem = torch.empty(B, fh.value * fw.value, self.arch["hidden_dim"], device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.odam_detr_debug_read(hd, B, _lib.ptr(l4), _lib.ptr(mem),
                                              ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                       "odam_detr_debug_read")
        return l4, mem

    # ---- DETR.postprocess (detr.py:96-159) ----------------------------------------------------------
    def postprocess_rows(self, out, img_size, intr_mat):
        """device part: [B,Q,16] rows (score, class, box, 3D centre, angle bin, dims) -> numpy"""
        H, W = out["_hw"]
        h = self._handle(H, W)
        B = out["pred_logits"].shape[0]
        rows = torch.empty(B, self.num_queries, 16, device=self.device, dtype=torch.float32)
        K9 = np.ascontiguousarray(np.asarray(intr_mat, np.float64)[:3, :3], np.float32)
        img_w, img_h = img_size
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().odam_detr_postprocess(
                h, _lib.ptr(out["pred_logits"]), _lib.ptr(out["pred_boxes"]), _lib.ptr(out["pred_angle"]),
                _lib.ptr(out["pred_offset"]), _lib.ptr(out["pred_size"]), _lib.ptr(out["pred_depth"]), ctypes.c_int(B),
                K9.ctypes.data_as(_lib.c_float_p), ctypes.c_float(img_w), ctypes.c_float(img_h), _lib.ptr(rows),
                ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "odam_detr_postprocess")
        return rows.cpu().numpy()

    @staticmethod
    def select(rows, threshold, nms_2d=True, n_bins=30):
        """host part: threshold + greedy NMS (odam_detr_select, C++) and packaging -- one frame's [Q,16] rows"""
        rows = np.ascontiguousarray(rows, np.float32)
        Q = rows.shape[0]
        keep = np.zeros(max(Q, 1), np.int32)
        n = ctypes.c_int()
        _lib.check(_lib.lib().odam_detr_select(rows.ctypes.data_as(_lib.c_float_p), ctypes.c_int(Q),
                                               ctypes.c_float(threshold), ctypes.c_int(int(bool(nms_2d))),
                                               keep.ctypes.data_as(_lib.c_int_p), ctypes.byref(n)), "odam_detr_select")
        r = rows[keep[:n.value]]
        return dict(bboxes=r[:, 2:6].reshape(-1, 2, 2).copy(), dimensions=r[:, 10:13].copy(),
                    angles=(r[:, 9] * np.float32(180 / n_bins)).astype(np.float32), translates=r[:, 6:9].copy(),
                    classes=r[:, 1].astype(np.int64), scores=r[:, 0].copy())

    def detect_resident(self, frames, img_size, intr_mat):
        """Forward + per-query post-processing over frames already resident on the device ([N,3,H,W] float32):
        all rows [N,Q,16] as one numpy array (see detect_resident_chunks for the streaming form)."""
        out = np.empty((frames.shape[0], self.num_queries, 16), np.float32)
        for c0, rows in self.detect_resident_chunks(frames, img_size, intr_mat, chunk=1 << 30):
            out[c0:c0 + rows.shape[0]] = rows
        return out

    @staticmethod
    def batch_for(n_frames):
        """frames per forward for a call over n_frames on three streams (bench.py --batch 0): measured on one box with cg.pin = 1
        (tests/native/sweep_shard_batch.sh, profiles/r6_shard_batch_sweep.txt) -- 125 frames: 42 per forward (three equal batches, one per
        stream) 135 ms against 141 with 32 (32 + 32 + 32 + 29: the fourth runs alone); 250: 42; 500 and 1000: 38 (whole rounds of 256 tiles in
        layer3 / layer4; 0.6-1.8 % over 32)"""
        return 42 if n_frames <= 300 else 38

    def detect_resident_chunks(self, frames, img_size, intr_mat, chunk=128, batch=None):
        """Forward + per-query post-processing over frames resident in HBM -- or, when `frames` is a (pinned) host
        tensor, uploaded batch by batch on a copy stream into two staging buffers per compute stream, so the
        PCIe transfer of batch n+1 runs under the kernels of batch n -- where (1) batches alternate between `n_streams` HIP streams, each with its own
        native handle/workspace, so the tile-quantisation tails, launch gaps and HBM-bound layers of one batch
        are filled by kernels of another (frames are independent); (2) the rows come back in chunks while the
        GPU keeps working: everything is enqueued first (forward + post-processing per batch, then per chunk an
        asynchronous copy into pinned host memory + an event), and the generator yields
        (first_frame, rows[n,Q,16]) as each chunk's event completes -- host-side thresholding / NMS of chunk i
        overlaps the kernels of chunks i+1.. .
        batch: frames per forward (<= max_batch, the default); "auto" = batch_for(number of frames)."""
        raw = frames.dtype == torch.uint8     # [N,h,w,3] frames as decoded: the input transform runs on the device
        if raw:
            from .transforms import target_size
            N, H, W = frames.shape[0], *target_size(frames.shape[2], frames.shape[1], *self.resize)
        else:
            N, _, H, W = frames.shape
        a = self.arch
        Q = a["num_queries"]
        dev = self.device
        L = _lib.lib()
        mb = self.max_batch                                   # what every buffer is sized for
        bb = mb if not batch else min(mb, self.batch_for(N) if batch == "auto" else int(batch))      # frames per forward of THIS call
        ns = self.n_streams
        handles = [self._handle(H, W, k) for k in range(ns)]
        chunk = max(bb, (chunk // bb) * bb)
        rows = torch.empty(N, Q, 16, device=dev, dtype=torch.float32)
        if getattr(self, "_pinned", None) is None or self._pinned.shape[0] < N:
            self._pinned = torch.empty(N, Q, 16, dtype=torch.float32, pin_memory=True)
        host = self._pinned[:N]
        if getattr(self, "_streams", None) is None or len(self._streams) != ns:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            mk = lambda n: torch.empty(mb, Q, n, device=dev, dtype=torch.float32)
            self._mscratch = [[mk(a["num_classes1"]), mk(4), mk(a["angle_bins"]), mk(2), mk(3), mk(1)] for _ in range(ns)]
        K9 = np.ascontiguousarray(np.asarray(intr_mat, np.float64)[:3, :3], np.float32)
        img_w, img_h = img_size
        main = torch.cuda.current_stream(dev)
        start = torch.cuda.Event()
        start.record(main)
        for st in self._streams:
            st.wait_event(start)          # inputs produced on the caller's stream are visible
        events = []
        nb = 0
        on_host = not frames.is_cuda
        if raw:
            if getattr(self, "_xform", None) is None or self._xform[0].shape[2:] != (H, W) or len(self._xform) != ns:
                self._xform = [torch.empty(mb, 3, H, W, device=dev, dtype=torch.float32) for _ in range(ns)]
        if on_host:
            st0 = getattr(self, "_stage", None)
            if st0 is None or st0[0][0].shape[1:] != frames.shape[1:] or st0[0][0].dtype != frames.dtype or len(st0) != ns:
                self._stage = [[torch.empty((mb,) + tuple(frames.shape[1:]), device=dev, dtype=frames.dtype)
                                for _ in range(2)] for _ in range(ns)]
                self._copy_stream = torch.cuda.Stream(device=dev)
            stage_free = [[None, None] for _ in range(ns)]   # event: the forward that last read this buffer is done
        with torch.cuda.device(dev):
            for c0 in range(0, N, chunk):
                c1 = min(N, c0 + chunk)
                used = set()
                for b0 in range(c0, c1, bb):
                    B = min(bb, c1 - b0)
                    k = nb % ns
                    nb += 1
                    used.add(k)
                    lg, bx, an, of, sz, dp = self._mscratch[k]
                    sp = ctypes.c_void_p(self._streams[k].cuda_stream)
                    src = frames[b0:b0 + B]
                    if on_host:
                        j = ((nb - 1) // ns) & 1
                        buf = self._stage[k][j][:B]
                        if stage_free[k][j] is not None:
                            self._copy_stream.wait_event(stage_free[k][j])
                        with torch.cuda.stream(self._copy_stream):
                            buf.copy_(src, non_blocking=True)
                            up = torch.cuda.Event()
                            up.record(self._copy_stream)
                        self._streams[k].wait_event(up)
                        src = buf
                    if raw:    # same stream as the forward that reads it: the buffer is free again when that is done
                        src = self.preprocess_u8(src, self._xform[k][:B], k, self._streams[k].cuda_stream)
                    _lib.check(L.odam_detr_forward(handles[k], _lib.ptr(src), ctypes.c_int(B), _lib.ptr(lg),
                                                   _lib.ptr(bx), _lib.ptr(an), _lib.ptr(of), _lib.ptr(sz), _lib.ptr(dp),
                                                   ctypes.c_void_p(0), sp), "odam_detr_forward")
                    _lib.check(L.odam_detr_postprocess(handles[k], _lib.ptr(lg), _lib.ptr(bx), _lib.ptr(an), _lib.ptr(of),
                                                       _lib.ptr(sz), _lib.ptr(dp), ctypes.c_int(B),
                                                       K9.ctypes.data_as(_lib.c_float_p), ctypes.c_float(img_w),
                                                       ctypes.c_float(img_h), _lib.ptr(rows[b0:b0 + B]), sp),
                               "odam_detr_postprocess")
                    if on_host:
                        stage_free[k][j] = torch.cuda.Event()
                        stage_free[k][j].record(self._streams[k])
                for k in used:               # the chunk's copy waits for every stream that produced part of it
                    e = torch.cuda.Event()
                    e.record(self._streams[k])
                    main.wait_event(e)
                host[c0:c1].copy_(rows[c0:c1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(main)
                events.append((c0, c1, ev))
        for c0, c1, ev in events:
            ev.synchronize()
            yield c0, host[c0:c1].numpy().copy()   # the pinned buffer is reused by the next call: rows leave as a copy

    def profile(self, H, W, on):
        _lib.check(_lib.lib().odam_detr_profile_enable(self._handle(H, W), ctypes.c_int(int(on))), "profile_enable")

    def profile_read(self, H, W):
        n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib().odam_detr_profile_read(self._handle(H, W), ctypes.byref(n), ctypes.byref(ms),
                                                     ctypes.byref(fl)), "profile_read")
        return n.value, ms.value, fl.value

    def profile_read_attention(self, H, W):
        n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib().odam_detr_profile_read_attention(self._handle(H, W), ctypes.byref(n), ctypes.byref(ms),
                                                               ctypes.byref(fl)), "profile_read_attention")
        return n.value, ms.value, fl.value

    def postprocess(self, out, img_size, threshold, intr_mat, nms_2d=True):
        rows = self.postprocess_rows(out, img_size, intr_mat)
        res = {k: [] for k in ("bboxes", "dimensions", "angles", "translates", "classes", "scores")}
        for b in range(rows.shape[0]):
            s = self.select(rows[b], threshold, nms_2d, self.arch["angle_bins"])
            for k in res:
                res[k].append(s[k])
        return res


def build(args):
    """src/models/detr.py:530-573 `build(cfg)` -> (model, criterion, postprocessors); inference only, so the
    training-time criterion / postprocessors are None.  Reads the same config keys."""
    g = (lambda k, d: args.get(k, d)) if isinstance(args, dict) else (lambda k, d: getattr(args, k, d))
    ds = g("dataset_file", "scan_net")
    num_classes = 18 if ds == "scan_net" else (91 if ds == "coco" else 20)
    if g("masks", False):
        raise _lib.OdamError("the segmentation head (masks) is not part of this path")
    det = Detector(pre_norm=g("pre_norm", False), position_embedding=g("position_embedding", "sine"), dilation=g("dilation", False),
                   backbone=g("backbone", "resnet50"), hidden_dim=g("hidden_dim", 256), nheads=g("nheads", 8),
                   dim_feedforward=g("dim_feedforward", 2048), enc_layers=g("enc_layers", 6),
                   dec_layers=g("dec_layers", 6), num_queries=g("num_queries", 100), num_classes=num_classes)
    return det, None, None
