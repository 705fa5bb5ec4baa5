"""Soak of OdamProcess.process_frames (one native call per frame, round 6) under a concurrent full-device detector load: a child process keeps
every CU busy with the detector's ring kernels while this process runs the association loop over N frames of the scene's detections; the
tracks that come out are compared with the quiet-device run of the same input (every row, bit for bit).  Prints one JSON line:
frames, identical, lost_launches (persistent launches abandoned and re-run through the launch sequence), ms per frame quiet / loaded.
   python3 tests/native/soak_assoc_fast.py [frames = 300] [load seconds = 40]"""
import json, os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import scene_weights
from odam_amd import associator, detector, sq, synth, weights
from odam_amd.processor import OdamProcess, detection_array
LOAD = """
import os, sys, time
sys.path.insert(0, %r)
import torch
from odam_amd import detector, weights
det = detector.Detector(max_batch=32, n_streams=1); det.load_state_dict(weights.make_state_dict())
img = torch.randn(32, 3, 800, 1066, device="cuda:0")
det(img); torch.cuda.synchronize()
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < %f: det(img)
torch.cuda.synchronize()
"""
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
load_s = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
dev = "cuda:0"
det = detector.Detector(max_batch=32, device=dev, n_streams=2); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
base = np.stack(list(synth.make_frames(32, seed=0)))
frames = torch.from_numpy(np.concatenate([np.clip(base.astype(np.int16) + k, 0, 255).astype(np.uint8) for k in range((N + 31) // 32)])[:N]).to(dev)
scene = synth.make_scene(1000, 30, seed=1, min_views=10, max_views=300)
per_frame = []
for c0, rows16 in det.detect_resident_chunks(frames, (synth.IMG_W, synth.IMG_H), synth.K_SCANNET, chunk=128):
    for i in range(rows16.shape[0]):
        sel = det.select(rows16[i], 0.6, True, 30)
        per_frame.append(detection_array({k: [v] for k, v in sel.items()}, 0, scene["img_names"][c0 + i], synth.IMG_W, synth.IMG_H))
det.close(); del frames; torch.cuda.empty_cache()
net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}, max_tracks=128, device=dev)
net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))
class _Det: device = dev
ids, T = [int(i) for i in scene["img_names"][:N]], [scene["T_wcs"][i] for i in range(N)]
def run():
    proc = OdamProcess(_Det(), net, None, None, fitter=sq.SqFitter(dev, 1))
    proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    proc.process_frames(ids, T, per_frame)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3, [t.copy() for t in proc.tracks]
run()
ms_quiet, ref = run()
lost0 = net.lost_launches
child = subprocess.Popen([sys.executable, "-c", LOAD % (ROOT, load_s)], stdout=subprocess.PIPE, text=True)
child.stdout.readline()
res = []
t_end = time.time() + load_s - 5
while time.time() < t_end:
    res.append(run())
child.wait()
same = all(len(tr) == len(ref) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(tr, ref)) for _, tr in res)
print(json.dumps({"frames_per_run": N, "runs_under_load": len(res), "identical_to_quiet_run": bool(same), "lost_launches_quiet": lost0,
                  "lost_launches_under_load": net.lost_launches - lost0, "ms_per_frame_quiet": ms_quiet,
                  "ms_per_frame_under_load": [round(m, 3) for m, _ in res], "persistent_given_up": bool(getattr(net, "persistent_given_up", False))}))
