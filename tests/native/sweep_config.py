"""One detector, interleaved rounds over SEVERAL values of one config key (same box, same process): forward time of 32 frames.
   python sweep_config.py <key> <v0,v1,...> [backbone dtype]"""
import os, sys, time, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib, detector, weights
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2].split(",")]
bb = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
det = detector.Detector(backbone=bb, max_batch=32, n_streams=1, dtype=dtype); det.load_state_dict(weights.make_state_dict(backbone=bb))
img = torch.randn(32, 3, 800, 1066, device="cuda:0")
res = {v: [] for v in vals}
for v in vals:
    _lib.set_config(key, v)
    for _ in range(2): det(img)
torch.cuda.synchronize()
for rnd in range(5):
    for v in vals:
        _lib.set_config(key, v)
        det(img); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): det(img)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 4 * 1e3)
for v in vals:
    print(f"{key}={v} {bb} {dtype}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  median {statistics.median(res[v]):.2f} ms per 32 frames", flush=True)
