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
