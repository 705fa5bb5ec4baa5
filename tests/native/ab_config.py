"""A/B of library switches on ONE box, interleaved in ONE process (cdna guide rule 24): python ab_config.py <key> <v0> <v1> [backbone dtype [frames]]
   forward of 32 (or `frames`) frames 800x1066, rounds alternate between the two values; prints the per-round times and the medians"""
import os, sys, time, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib, detector, weights
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bb = sys.argv[4] if len(sys.argv) > 4 else "resnet50"
dtype = sys.argv[5] if len(sys.argv) > 5 else "fp32"
NB = int(sys.argv[6]) if len(sys.argv) > 6 else 32
det = detector.Detector(backbone=bb, max_batch=NB, n_streams=1, dtype=dtype); det.load_state_dict(weights.make_state_dict(backbone=bb))
img = torch.randn(NB, 3, 800, 1066, device="cuda:0")
res = {v0: [], v1: []}
for v in (v0, v1):
    _lib.set_config(key, v)
    for _ in range(2): det(img)
torch.cuda.synchronize()
for rnd in range(6):
    for v in (v0, v1):
        _lib.set_config(key, v)
        det(img); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): det(img)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 4 * 1e3)
import hashlib
def _dig(v):
    _lib.set_config(key, v)
    out = det(img); torch.cuda.synchronize()
    h = hashlib.sha1()
    for k in sorted(out):
        if hasattr(out[k], "cpu"): h.update(out[k].float().cpu().numpy().tobytes())
    return h.hexdigest()[:16]
print("outputs:", {v: _dig(v) for v in (v0, v1)}, flush=True)
for v in (v0, v1):
    print(f"{key}={v} {bb} {dtype}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  median {statistics.median(res[v]):.2f} ms per {NB} frames", flush=True)
