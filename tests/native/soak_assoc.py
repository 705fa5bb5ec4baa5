"""Soak of the association call under a concurrent full-device detector load.

A child process keeps every CU busy with the detector's one-workgroup-per-CU ring kernels (B = 32 forwards back to back)
while this process runs N association frames (default 300) through Associator.__call__ -- the persistent matching launch,
the lost-launch check and, where a launch is lost, the same-process re-run through the launch sequence.  Every frame's
matches are compared with the quiet-device result of the same input.  Prints one JSON line:
  frames, failures (exceptions or wrong matches: must be 0), lost_launches (re-run frames), ms per frame quiet / loaded.
Usage: python3 tests/native/soak_assoc.py [frames] [load_seconds]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)

LOAD = """
import os, sys, time
sys.path.insert(0, %r)
import torch
from odam_amd import detector, weights
det = detector.Detector(max_batch=32, n_streams=1); det.load_state_dict(weights.make_state_dict())
img = torch.randn(32, 3, 800, 1066, device="cuda:0")
det(img); torch.cuda.synchronize()
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < %f: det(img)
torch.cuda.synchronize()
"""


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    load_s = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    import numpy as np
    import torch
    from odam_amd import associator, weights
    z = np.load(os.path.join(ROOT, "tests/golden/assoc.npz"))
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    a = associator.Associator(cfg, max_tracks=64)
    a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    cases = []
    for c in range(int(z["n_cases"])):
        Zr = z[f"c{c}_Z"]
        T, n = Zr.shape[0] - 1, Zr.shape[1] - 1
        data = {"tracks": torch.from_numpy(z[f"c{c}_tracks"]).cuda(), "detections": torch.from_numpy(z[f"c{c}_det"]).cuda(),
                "valid_list": [(T, n)]}
        quiet = a(data, 0.1, eval_only=True)
        cases.append((data, quiet["matches"][0], quiet["pred"][0][0].cpu().numpy()))
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(100):
        a(cases[i % len(cases)][0], 0.1, eval_only=True)
    ms_quiet = (time.time() - t0) / 100 * 1e3
    lost_quiet = a.lost_launches

    child = subprocess.Popen([sys.executable, "-c", LOAD % (ROOT, load_s)], stdout=subprocess.PIPE, text=True)
    child.stdout.readline()      # "ready": the load is running
    failures, worst = 0, 0.0
    t0 = time.time()
    for i in range(frames):
        data, matches, Zq = cases[i % len(cases)]
        try:
            out = a(data, 0.1, eval_only=True)
            Z = out["pred"][0][0].cpu().numpy()
            if not np.isfinite(Z).all() or not np.array_equal(out["matches"][0], matches):
                failures += 1
            worst = max(worst, float(np.abs(np.exp(Z) - np.exp(Zq)).max()))
        except Exception as e:      # noqa: BLE001 -- the soak counts, it does not stop
            failures += 1
            print("frame", i, "raised", repr(e), file=sys.stderr)
    ms_loaded = (time.time() - t0) / frames * 1e3
    still_loaded = child.poll() is None
    child.wait()
    print(json.dumps({"frames": frames, "failures": failures, "lost_launches": a.lost_launches - lost_quiet,
                      "lost_launches_quiet": lost_quiet, "ms_per_frame_quiet": round(ms_quiet, 3),
                      "ms_per_frame_under_load": round(ms_loaded, 3), "max_abs_dP_vs_quiet": worst,
                      "load_ran_throughout": still_loaded}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
