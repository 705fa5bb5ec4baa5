"""two launches of the fit kernel for the PMC passes of run_pmc.sh:  prof_sq.py [objects views]   (default 64 x 256; configs[4]: 500 256)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
views = int(sys.argv[2]) if len(sys.argv) > 2 else 256
f = sq.SqFitter("cuda:0", 200)
base = [synth.make_sq_problem(views, s) for s in range(min(n_obj, 64))]
probs = [base[i % len(base)] for i in range(n_obj)]
p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs]); tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
ms = []
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f.fit(p0, [p["class_id"] for p in probs], [views] * n_obj, P, tgt, mask); e1.record()
    torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
print("sq fit ok: %d objects x %d views, ms per launch (events around the host call) %s" % (n_obj, views, ["%.2f" % m for m in ms]))
