"""matching GNN: the row-partitioned kernel (odam_config assoc.persist = 2: one device-wide barrier per layer, the rest XCD-local) against
the device-wide persistent kernel (1): assignment matrices bit for bit on the reference-run cases, launch time by stage stamps"""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import associator, weights, _lib
z = np.load(os.path.join(ROOT, "tests/golden/assoc.npz"))
cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
a = associator.Associator(cfg, max_tracks=128, device="cuda:0"); a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
L = _lib.lib()
rng = np.random.default_rng(0)
cases = [(z[f"c{c}_tracks"], z[f"c{c}_det"], z[f"c{c}_Z"].shape[0] - 1, z[f"c{c}_Z"].shape[1] - 1) for c in range(int(z["n_cases"]))]
for T in (1, 7, 8, 9, 63, 64, 65, 100, 127):          # more row counts: 8 XCDs x ceil((T + 30) / 8) rows, several row blocks per XCD
    cases.append((rng.standard_normal((T, 79, 100)).astype(np.float32), rng.standard_normal((79, 30)).astype(np.float32), T, int(rng.integers(1, 31))))
for tr, de, T, n in cases:
    out = {}
    for v in (1, 2):
        _lib.set_config("assoc.persist", v)
        Z = a.assignment_checked(torch.from_numpy(tr), torch.from_numpy(de), T, n).cpu().numpy()
        buf = (ctypes.c_ulonglong * 64)()
        L.odam_assoc_stage_stamps(a._handle(), 1, None, 0)
        a.assignment(torch.from_numpy(tr), torch.from_numpy(de), T, n); a.assignment(torch.from_numpy(tr), torch.from_numpy(de), T, n)
        L.odam_assoc_stage_stamps(a._handle(), 0, buf, 64)
        t = np.array(buf[:42], dtype=np.int64)
        out[v] = (Z, (t[41] - t[0]) / 100.0, np.diff(t)[:5] / 100.0)
    same = np.array_equal(out[1][0].view(np.uint32), out[2][0].view(np.uint32))
    print(f"T={T:4d} n={n:2d}: bit-identical {same}   launch {out[1][1]:.1f} -> {out[2][1]:.1f} us   first layer stages {np.round(out[1][2], 1)} -> {np.round(out[2][2], 1)}   lost {a.lost_launches}", flush=True)
_lib.set_config("assoc.persist", 2)
