"""prints a digest of the fitted parameters / losses / points of a fixed problem set (run under different
ODAM_SQ_SPLIT / ODAM_SQ_SPLIT_WAIT settings; the digests must agree)"""
import hashlib, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 60)
h = hashlib.sha256()
for vc in ([298], [64, 65, 200, 130], [300, 10, 90, 128, 257, 70, 33, 64, 199]):
    probs = [synth.make_sq_problem(F, 7 + i) for i, F in enumerate(vc)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs])
    tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    out = f.fit(p0, [p["class_id"] for p in probs], vc, P, tgt, mask, n_iters=60, want_points=True, want_loss=True)
    for k in ("params", "points", "loss"):
        h.update(np.ascontiguousarray(out[k].cpu().numpy()).tobytes())
print("digest", h.hexdigest())
