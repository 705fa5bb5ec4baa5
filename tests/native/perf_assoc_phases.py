"""wall clock per phase of a frame of the association loop (perf_assoc_loop.py's scene): method wrappers with perf_counter, no profiler"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scene_weights
from odam_amd import associator, detector, sq, synth, weights, processor
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
net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}, max_tracks=128, device=dev)
net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))
T = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name); key = key or name
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
class _Det: device = dev
def run():
    proc = OdamProcess(_Det(), net, None, None, fitter=sq.SqFitter(dev, 1))
    proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
    for n in ("_prepare_tracks", "_collater", "_attach_to_tracks", "_windows", "_track_rows"): wrap(proc, n)
    ids, Tw = scene["img_names"], scene["T_wcs"]
    t0 = time.perf_counter()
    for i in range(N):
        proc.process_frame(None, int(ids[i]), Tw[i], detections=per_frame[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
run()
wrap(net, "assignment"); wrap(associator, "hungarian_matching")
st = torch.cuda.current_stream(torch.device(dev))
_sync = torch.cuda.Stream.synchronize
def sync(self):
    t0 = time.perf_counter(); _sync(self); T["synchronize"] = T.get("synchronize", 0.0) + time.perf_counter() - t0
torch.cuda.Stream.synchronize = sync
wrap(net, "__call__", "associator.__call__")
fitter_cls = sq.SqFitter
wrap(fitter_cls, "project_extents") if False else None
T.clear()
ms = run()
print(f"{ms:.3f} ms per frame")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"  {k:28s} {v / N * 1e3:7.3f} ms per frame")
