"""exact duplicate tracks must get bit-identical rows of the assignment matrix (the Hungarian tie-break then goes by index, as in the reference)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from odam_amd import _lib, associator, weights
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scene_weights
DEV = "cuda:0"
z = np.load(os.path.join(ROOT, "tests/golden/assoc.npz"))
cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
for scene in (False, True):
    sd = scene_weights.make_scene_associator_state_dict(2, 8, seed=0) if scene else weights.make_associator_state_dict(2, 8, seed=0)
    for env in ("1", "0"):
        _lib.set_config("assoc.persist", int(env))      # read when the handle is finalised
        a = associator.Associator(cfg, max_tracks=64, device=DEV); a.load_state_dict(sd); a._handle()
        base = torch.from_numpy(z["c3_tracks"]); de = torch.from_numpy(z["c3_det"])
        T = 40
        tr = base[:T].clone()
        for dst, src in ((27, 6), (7, 6), (33, 2), (18, 17)):
            tr[dst] = tr[src]
        Z = a.assignment(tr.to(DEV), de.to(DEV), T, 12).cpu().numpy()
        eq = [np.array_equal(Z[d], Z[s]) for d, s in ((27, 6), (7, 6), (33, 2), (18, 17))]
        dd = [np.abs(np.exp(Z[d]) - np.exp(Z[s])).max() for d, s in ((27, 6), (7, 6), (33, 2), (18, 17))]
        print(f"scene={scene} persist={env}: duplicate rows bit-identical: {eq}  max |dP| {dd}")
        a.close()
