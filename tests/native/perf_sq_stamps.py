"""per-phase time of one Adam step (build with -DSQ_STAMPS into tests/native/_abl/libstamps.so)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "tests/native/_abl/libstamps.so")
from odam_amd import sq, synth
f = sq.SqFitter("cuda:0", 200)
names = ["A shape", "B1 table", "B2 levels", "C cdf", "D points", "E exchange", "F tree/adam", "E own views"]
for F in (16, 128, 298):
    probs = [synth.make_sq_problem(F, 3)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs]); tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    out = f.fit(p0, [probs[0]["class_id"]], [F], P, tgt, mask, n_iters=200, want_points=True)
    st = out["points"].cpu().numpy().reshape(-1)[:8] / 100.0 / 200   # us per step
    print(f"F={F}: " + "  ".join(f"{n} {v:5.1f}" for n, v in zip(names, st)) + f"   total {st.sum():.1f} us/step", flush=True)
