"""ms per frame of the association loop, step-by-step path vs OdamProcess.process_frames (round 6), same detections:
   python tests/native/perf_assoc_fast.py [frames]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scene_weights
from odam_amd import associator, detector, sq, synth, weights
from odam_amd.processor import OdamProcess, detection_array
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
det = detector.Detector(max_batch=32, device=dev, n_streams=2); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
base = np.stack(list(synth.make_frames(32, seed=0)))
frames = torch.from_numpy(np.concatenate([np.clip(base.astype(np.int16) + k, 0, 255).astype(np.uint8) for k in range((N + 31) // 32)])[:N]).to(dev)
scene = synth.make_scene(1000, 30, seed=1, min_views=10, max_views=300)
per_frame = []
for c0, rows16 in det.detect_resident_chunks(frames, (synth.IMG_W, synth.IMG_H), synth.K_SCANNET, chunk=128):
    for i in range(rows16.shape[0]):
        sel = det.select(rows16[i], 0.6, True, 30)
        per_frame.append(detection_array({k: [v] for k, v in sel.items()}, 0, scene["img_names"][c0 + i], synth.IMG_W, synth.IMG_H))
det.close()
if os.environ.get("ASSOC_CG_FORCE"):      # experiment: one tile shape for every contraction of the association network (odam_config cg.force)
    from odam_amd import _lib
    _lib.set_config("cg.force", int(os.environ["ASSOC_CG_FORCE"]))
net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}, max_tracks=128, device=dev)
net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))
class _Det: device = dev
ids, T = [int(i) for i in scene["img_names"][:N]], [scene["T_wcs"][i] for i in range(N)]
def run(mode):
    proc = OdamProcess(_Det(), net, None, None, fitter=sq.SqFitter(dev, 1))
    proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "host":
        for i in range(N):
            proc._process_frame_host(None, ids[i], T[i], per_frame[i])
    elif mode == "single":
        for i in range(N):
            proc.process_frame(None, ids[i], T[i], detections=per_frame[i])
    else:
        proc.process_frames(ids, T, per_frame)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    return dt, [t.copy() for t in proc.tracks]
def _attach_numpy(self, Z, cur):
    """the attach step as numpy + scipy computed it before odam_assoc_attach (A/B: ATTACH_AB=1 adds a 'batched-np' line per repetition)"""
    from scipy.optimize import linear_sum_assignment
    n_det = cur["det32T"].shape[1]
    score_mat = Z.exp().numpy()
    scores = score_mat[:-1, :-1]
    r_, c_ = linear_sum_assignment(1 - scores)
    ok = scores[r_, c_] > self.match_threshold
    match = np.full(n_det, -1, np.int32)
    match[c_[ok]] = r_[ok]
    keep = ~(score_mat[match, np.arange(n_det)] < self.score_threshold)
    picked = np.flatnonzero(keep)
    mk = match[picked]
    new = mk == -1
    n_tracks = self._n_tracks
    ids_a = mk.astype(np.int64)
    ids_a[new] = n_tracks + np.arange(int(new.sum()))
    n_tracks += int(new.sum())
    ids_ = ids_a.tolist()
    if ids_:
        sel = cur["rows82"][picked]
        self._pending.append((ids_, sel)); self._pending_app = (ids_, sel); self._n_tracks = n_tracks; self._win_meta_stale = True
_native_attach = OdamProcess._attach_fast
ref = None
modes = ("host", "single", "batched") + (("batched-np", "batched") if os.environ.get("ATTACH_AB") else ())
for rep in range(2 if not os.environ.get("ATTACH_AB") else 3):
    for mode in modes:
        OdamProcess._attach_fast = _attach_numpy if mode == "batched-np" else _native_attach
        ms, tr = run("batched" if mode == "batched-np" else mode)
        if ref is None:
            ref = tr
        same = len(tr) == len(ref) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(tr, ref))
        print(f"{mode:8s} {ms:.3f} ms per frame, {len(tr)} tracks, identical to the step-by-step path: {same}", flush=True)
OdamProcess._attach_fast = _native_attach
print("lost launches", net.lost_launches)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run("batched"); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
