"""us/step of the fit for one object with F views, lanes-per-view forced (ODAM_SQ_S) -- run once per S"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
res = []
for F in (10, 19, 32, 38, 64, 75, 100, 128, 150, 200, 256, 298, 400, 512, 1024):
    p = synth.make_sq_problem(F, 5)
    p0 = sq.init_params(p["translate"], p["angle"], p["dims"])[None]
    t, m = sq.lines_to_targets(p["bbox_lines"])
    P = p["P"].astype(np.float32).reshape(-1, 12)
    args = (p0, [p["class_id"]], [F], P, t, m)
    f.fit(*args, n_iters=200); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): f.fit(*args, n_iters=200)
    torch.cuda.synchronize(); res.append((F, (time.perf_counter() - t0) / 2 / 200 * 1e6))
print("S=%s " % os.environ.get("ODAM_SQ_S", "auto") + " ".join(f"{F}:{us:.1f}" for F, us in res), flush=True)
