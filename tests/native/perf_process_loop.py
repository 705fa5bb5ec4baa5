"""per-frame cost of the drop-in loop: detector at B=1 (the reference calls it once per frame), and
OdamProcess.process_frame with the association network on precomputed detections"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import detector, weights, synth, associator, processor, transforms
dev = "cuda:0"
det = detector.Detector(max_batch=1, n_streams=1); det.load_state_dict(weights.make_state_dict())
img = torch.randn(1, 3, 800, 1066, device=dev)
for _ in range(3): det(img)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): det(img)
torch.cuda.synchronize(); print(f"detector B=1 forward: {(time.perf_counter()-t0)/20*1e3:.2f} ms", flush=True)
for _ in range(3):
    out = det(img); det.postprocess(out, (1296, 968), 0.6, synth.K_SCANNET)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    out = det(img); det.postprocess(out, (1296, 968), 0.6, synth.K_SCANNET)
print(f"detector B=1 forward + postprocess (sync per frame): {(time.perf_counter()-t0)/20*1e3:.2f} ms", flush=True)

z = np.load(os.path.join(ROOT, "tests/golden/process_tracks_net.npz"), allow_pickle=True)
print(sorted(z.files)[:12])

# ---- the association replay: process_frame with injected detections, real association network ----------
from odam_amd import sq
from odam_amd.processor import OdamProcess
import cProfile, pstats
z = np.load(os.path.join(ROOT, "tests/golden/process_tracks.npz"), allow_pickle=True)
net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                             "sinkhorn_iterations": 100}, max_tracks=64, device=dev)
net.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
class _Det: device = dev
def replay():
    proc = OdamProcess(_Det(), net, None, None, score_threshold=0.2, fitter=sq.SqFitter(dev, 1))
    proc.init_sequence(z["K"], 480, 640)
    for i, fid in enumerate(z["img_names"]):
        proc.process_frame(None, int(fid), z["scene_T_wcs"][i], detections=[list(r) for r in z[f"det{int(fid)}"]])
    return proc
replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): p = replay()
torch.cuda.synchronize(); n = len(z["img_names"])
print(f"association replay: {(time.perf_counter()-t0)/3/n*1e3:.2f} ms per frame ({n} frames, {len(p.tracks)} tracks)", flush=True)
pr = cProfile.Profile(); pr.enable(); replay(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
import timeit
x = np.random.rand(1000, 3)
print("np.max [1000,3] axis 0:", timeit.timeit(lambda: np.max(x, axis=0), number=200) / 200 * 1e6, "us")
print("torch.ones(5,100,79)*-1:", timeit.timeit(lambda: torch.ones((5, 100, 79)) * -1, number=200) / 200 * 1e6, "us")
print("torch threads", torch.get_num_threads())
