"""where the SQ phase of a bench step goes: host constraint building vs the fit launch vs result handling"""
import os, sys, time, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import multi_view, sq, synth
fitter = sq.SqFitter("cuda:0", 200)
scene = synth.make_scene(1000, 30, seed=1, min_views=10, max_views=300)
K = synth.K_SCANNET
def run():
    return multi_view.optim_process(scene["tracks"], scene["img_names"], scene["T_wcs"], scene["P_cws"], synth.IMG_H, synth.IMG_W,
                                    K, "super_quadric", True, 200, 10, fitter=fitter, return_params=True)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize(); print(f"optim_process: {(time.perf_counter()-t0)/5*1e3:.1f} ms per pass", flush=True)
orig = fitter.fit
acc = []
def timed(*a, **k):
    torch.cuda.synchronize(); s = time.perf_counter(); o = orig(*a, **k); torch.cuda.synchronize(); acc.append(time.perf_counter() - s); return o
fitter.fit = timed
for _ in range(5): run()
print(f"fitter.fit (upload + kernel + sync): {np.mean(acc)*1e3:.1f} ms", flush=True)
fitter.fit = orig
pr = cProfile.Profile(); pr.enable()
for _ in range(3): run()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
