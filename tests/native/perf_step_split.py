"""wall clock of the pieces of one headline step (bench.py step(), one GPU): detection loop (GPU phase by events vs wall), packing, the two
fit passes"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import detector as det_mod, multi_view, parallel, sq, synth, weights
from odam_amd.processor import detection_array
dev = "cuda:0"; N = 1000
det = det_mod.Detector(max_batch=32, device=dev, n_streams=3); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
base = np.stack(list(synth.make_frames(32, seed=0)))
frames = torch.from_numpy(np.concatenate([base] * 32)[:N]).to(dev)
scene = synth.make_scene(N, 30, seed=1, min_views=10, max_views=300)
fitter = sq.SqFitter(dev, 200); ids = scene["img_names"]; K = synth.K_SCANNET
def step(T):
    t0 = time.perf_counter(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    per_frame = []; t_sel = 0.0; t_wait = 0.0; tw = time.perf_counter()
    for c0, rows16 in det.detect_resident_chunks(frames, (synth.IMG_W, synth.IMG_H), K, chunk=128):
        t_wait += time.perf_counter() - tw; ts = time.perf_counter()
        for i in range(rows16.shape[0]):
            sel = det.select(rows16[i], 0.6, True, 30)
            per_frame.append(detection_array({k: [v] for k, v in sel.items()}, 0, ids[c0 + i], synth.IMG_W, synth.IMG_H))
        t_sel += time.perf_counter() - ts; tw = time.perf_counter()
    e1.record(); t1 = time.perf_counter()
    blk, cnt = parallel.pack_detections(per_frame); t2 = time.perf_counter()
    for _ in range(2):
        out = multi_view.optim_process(scene["tracks"], ids, scene["T_wcs"], scene["P_cws"], synth.IMG_H, synth.IMG_W, K, "super_quadric", True, 200, 10, fitter=fitter, return_params=True)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    T.append((1e3 * (t1 - t0), e0.elapsed_time(e1), 1e3 * t_wait, 1e3 * t_sel, 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)))
T = []
step(T); T.clear()
for _ in range(3): step(T)
a = np.mean(T, axis=0)
print("detection loop wall %.1f ms (GPU phase by events %.1f; waiting for chunks %.1f, host select + rows %.1f)  pack %.1f  two fit passes %.1f  step %.1f" % tuple(a))
