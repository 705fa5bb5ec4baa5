"""the fit launch of the bench's 30-object scene (10 ... 300 views), kernel time by HIP events, for the library named by ODAM_AMD_LIB:
   python perf_sq_fit30.py [every]  -> ms per launch (median of 7), us per Adam step; every = 8: objects 0, 8, 16, 24 only (what rank 0 of eight fits)"""
import os, sys, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import multi_view, sq, synth
fitter = sq.SqFitter("cuda:0", 200)
scene = synth.make_scene(1000, 30, seed=1, min_views=10, max_views=300)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scene["tracks"] = scene["tracks"][0::every]
rec = []
orig = fitter.fit
def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(*a, **k); e1.record(); rec.append((e0, e1)); return out
fitter.fit = timed
for _ in range(8):
    multi_view.optim_process(scene["tracks"], scene["img_names"], scene["T_wcs"], scene["P_cws"], synth.IMG_H, synth.IMG_W, synth.K_SCANNET,
                             "super_quadric", True, 200, 10, fitter=fitter, return_params=True)
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in rec[1:]]
print(f"{os.path.basename(os.environ.get('ODAM_AMD_LIB', 'libodam_amd.so'))} split={os.environ.get('ODAM_CONFIG', 'default')}: fit launch of {len(scene['tracks'])} objects {statistics.median(ms):.2f} ms (min {min(ms):.2f}) = {statistics.median(ms) * 5:.1f} us per Adam step")
