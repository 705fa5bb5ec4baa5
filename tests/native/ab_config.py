"""A/B of library switches on ONE box, interleaved in ONE process (cdna guide rule 24): python ab_config.py <key> <v0> <v1> [backbone dtype]
   forward of 32 frames 800x1066, rounds alternate between the two values; prints the per-round times and the medians"""
import os, sys, time, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib, detector, weights
key, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bb = sys.argv[4] if len(sys.argv) > 4 else "resnet50"
dtype = sys.argv[5] if len(sys.argv) > 5 else "fp32"
det = detector.Detector(backbone=bb, max_batch=32, n_streams=1, dtype=dtype); det.load_state_dict(weights.make_state_dict(backbone=bb))
img = torch.randn(32, 3, 800, 1066, device="cuda:0")
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
for v in (v0, v1):
    print(f"{key}={v} {bb} {dtype}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  median {statistics.median(res[v]):.2f} ms per 32 frames", flush=True)
