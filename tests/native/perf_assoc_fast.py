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
ref = None
for rep in range(2):
    for mode in ("host", "single", "batched"):
        ms, tr = run(mode)
        if ref is None:
            ref = tr
        same = len(tr) == len(ref) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(tr, ref))
        print(f"{mode:8s} {ms:.3f} ms per frame, {len(tr)} tracks, identical to the step-by-step path: {same}", flush=True)
print("lost launches", net.lost_launches)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); run("batched"); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
