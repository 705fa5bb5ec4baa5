"""forward time of 32 frames with the library named by ODAM_AMD_LIB (or the in-tree one), and a digest of the outputs:
   python fwd_time.py [backbone dtype B reps]   -> one line: "<median ms> <min ms> <sha1 of the six outputs>" """
import os, sys, time, statistics, hashlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
bb = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
det = detector.Detector(backbone=bb, max_batch=B, n_streams=1, dtype=dtype); det.load_state_dict(weights.make_state_dict(backbone=bb, scene=True))
torch.manual_seed(0)
img = torch.randn(B, 3, 800, 1066, device="cuda:0")
for _ in range(2): o = det(img)
torch.cuda.synchronize()
h = hashlib.sha1()
for k in sorted(o):
    if torch.is_tensor(o[k]): h.update(o[k].float().cpu().numpy().tobytes())
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    for _ in range(3): det(img)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
print(f"{statistics.median(ts):.3f} {min(ts):.3f} {h.hexdigest()[:16]}")
