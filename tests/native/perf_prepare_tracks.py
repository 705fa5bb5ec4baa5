import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq
from odam_amd.processor import OdamProcess
z = np.load(os.path.join(ROOT, "tests/golden/process_tracks.npz"), allow_pickle=True)
class _Det: device = "cuda:0"
proc = OdamProcess(_Det(), None, None, None, score_threshold=0.2, fitter=sq.SqFitter("cuda:0", 1))
proc.init_sequence(z["K"], 480, 640)
base = [z[f"det{int(f)}"] for f in z["img_names"][:3]]
T_wc = z["scene_T_wcs"][0]
tr = proc._init_tracks(np.concatenate(base)[:8], T_wc)
for n in (5, 20, 40, 80):
    proc.tracks = [tr[i % len(tr)].repeat(1 + i % 7, 0) for i in range(n)]
    proc._prepare_tracks(T_wc); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): proc._prepare_tracks(T_wc)
    print(f"{n} tracks: _prepare_tracks {(time.perf_counter()-t0)/20*1e3:.2f} ms", flush=True)
import cProfile, pstats
proc.tracks = [tr[i % len(tr)].repeat(1 + i % 7, 0) for i in range(40)]
pr = cProfile.Profile(); pr.enable()
for _ in range(20): proc._prepare_tracks(T_wc)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
