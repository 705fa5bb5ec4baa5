import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
probs = [synth.make_sq_problem(256, s) for s in range(64)]
p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs]); tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
for _ in range(2): f.fit(p0, [p["class_id"] for p in probs], [256] * 64, P, tgt, mask)
torch.cuda.synchronize(); print("sq fit ok")
