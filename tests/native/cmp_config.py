"""One detector, one batch, two values of a config key: largest absolute difference of every output (0 = the two kernels agree bit
for bit), and the forward time under each (usage: cmp_config.py cg.fuse 2 3 [B])"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights, _lib
key, v1, v2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
det = detector.Detector(max_batch=B, n_streams=1); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
torch.manual_seed(0)
img = torch.randn(B, 3, 800, 1066, device="cuda:0")
outs = {}
for v in (v1, v2, v1, v2):
    _lib.set_config(key, v)
    o = det(img); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): o = det(img)
    torch.cuda.synchronize()
    print(f"{key}={v}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per forward of {B}", flush=True)
    outs[v] = {k: t.clone() for k, t in o.items() if torch.is_tensor(t)}
for k in outs[v1]:
    d = (outs[v1][k].float() - outs[v2][k].float()).abs().max().item()
    print(f"  {k}: max |{key}={v1} - {key}={v2}| = {d:.3e}   (range {outs[v1][k].float().abs().max().item():.3e})")
