import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import associator, weights
z = np.load(os.path.join(ROOT, "tests/golden/assoc.npz"))
a = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}, max_tracks=64)
a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
for c in range(int(z["n_cases"])):
    tr, de, Zr = torch.from_numpy(z[f"c{c}_tracks"]).cuda(), torch.from_numpy(z[f"c{c}_det"]).cuda(), z[f"c{c}_Z"]
    T, n = Zr.shape[0] - 1, Zr.shape[1] - 1
    Z = a.assignment(tr, de, T, n).cpu().numpy()
    print(f"case {c} T={T} n={n}: max|dP| {np.abs(np.exp(Z)-np.exp(Zr)).max():.2e}  max|dZ| (Z>-6) {np.abs(Z[Zr>-6]-Zr[Zr>-6]).max():.2e}", end="  ")
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): a.assignment(tr, de, T, n)
    torch.cuda.synchronize(); print(f"{(time.time()-t0)/50*1e3:.3f} ms per frame (GNN + Sinkhorn, incl. host launch)")
# stage stamps of the persistent matching kernel (last case)
import ctypes
from odam_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 64)()
L.odam_assoc_stage_stamps(a._handle(), 1, None, 0)
a.assignment(tr, de, T, n); a.assignment(tr, de, T, n)
L.odam_assoc_stage_stamps(a._handle(), 0, buf, 64)
t = np.array(buf[:42], dtype=np.int64)
d = np.diff(t) / 100.0
print("stage us (8 x [qkv attn merge m0 m2], final):", " ".join(f"{x:.1f}" for x in d), "| total", (t[41] - t[0]) / 100.0)
