"""fit launch time with the view split on/off: run as  ODAM_SQ_SPLIT=1|2|4 python3 perf_sq_split.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
for n_obj, views in ((1, 298), (4, 298), (4, 128), (30, None), (60, 256)):
    if views is None:
        scene = synth.make_scene(1000, 30, seed=1, min_views=10, max_views=300)
        vc = [min(len(t), 300) for t in scene["tracks"]]
    else:
        vc = [views] * n_obj
    probs = [synth.make_sq_problem(F, 40 + i) for i, F in enumerate(vc)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs])
    tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    cls = [p["class_id"] for p in probs]
    out = f.fit(p0, cls, vc, P, tgt, mask, n_iters=200)["params"].cpu().numpy()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f.fit(p0, cls, vc, P, tgt, mask, n_iters=200)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"split={os.environ.get('ODAM_SQ_SPLIT','4')} objects {len(vc):3d} max views {max(vc):4d}: {dt*1e3:7.2f} ms per 200-step fit ({dt/200*1e6:6.1f} us/step)  checksum {float(np.abs(out).sum()):.6f}", flush=True)
