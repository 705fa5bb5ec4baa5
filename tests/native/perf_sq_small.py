import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
for F in (16, 256):
    probs = [synth.make_sq_problem(F, s) for s in range(32)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs]); tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    dP, dt_, dm = torch.from_numpy(P).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(mask).cuda()
    cls = [p["class_id"] for p in probs]
    f.fit(p0, cls, [F] * 32, dP, dt_, dm, want_points=False); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): f.fit(p0, cls, [F] * 32, dP, dt_, dm, want_points=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print(f"F={F:5d}: {dt/200*1e6:6.1f} us/step", flush=True)
