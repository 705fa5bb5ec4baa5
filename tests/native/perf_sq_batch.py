"""fit-kernel time over object counts (the bench scene: 30 objects 10-300 views; config 4: 50; config 5: 500 x 256)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
rs = np.random.RandomState(0)
for n, views in ((30, None), (50, None), (128, 64), (500, 256)):
    vc = [int(v) for v in (rs.randint(10, 300, n) if views is None else [views] * n)]
    probs = [synth.make_sq_problem(F, s) for s, F in enumerate(vc)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = torch.from_numpy(np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs])).cuda()
    tgt = torch.from_numpy(np.concatenate([t for t, _ in tm])).cuda(); mask = torch.from_numpy(np.concatenate([m for _, m in tm])).cuda()
    cls = [p["class_id"] for p in probs]
    out = f.fit(p0, cls, vc, P, tgt, mask, want_points=False)["params"]; torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): f.fit(p0, cls, vc, P, tgt, mask, want_points=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"{n:4d} objects ({'10-300' if views is None else views} views): {dt*1e3:8.2f} ms per launch  digest {float(out.double().sum()):.10f}", flush=True)
