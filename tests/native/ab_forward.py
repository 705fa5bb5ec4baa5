"""A/B of two builds of the library on the same box: python ab_forward.py  (ODAM_AMD_LIB picks the build; fp32 R50 forward, B = 32)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
bb = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
sd = weights.make_state_dict(backbone=bb)
B = 32
det = detector.Detector(backbone=bb, max_batch=B, n_streams=1, dtype=dtype); det.load_state_dict(sd)
img = torch.randn(B, 3, 800, 1066, device="cuda:0")
for _ in range(3): det(img)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.time()
    for _ in range(5): det(img)
    torch.cuda.synchronize(); ts.append((time.time() - t0) / 5 * 1e3)
print(f"{os.environ.get('ODAM_AMD_LIB', 'default')[-20:]} {bb} {dtype}: " + " ".join(f"{t:.2f}" for t in ts) + " ms/batch", flush=True)
