"""Forward time per frame against the frames per forward: the ring kernels run one 256-row tile per CU at a time, so a layer's
time goes in steps of whole rounds of 256 tiles (usage: perf_batch_sweep.py fp32|bf16 resnet50|resnet101 B [B ...])"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
DT, BB = sys.argv[1], sys.argv[2]
sd = weights.make_state_dict(backbone=BB)
for B in [int(b) for b in sys.argv[3:]]:
    det = detector.Detector(backbone=BB, max_batch=B, n_streams=1, dtype=DT); det.load_state_dict(sd)
    img = torch.randn(B, 3, 800, 1066, device="cuda:0")
    det(img); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); det(img); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print(f"{DT} {BB} B={B:3d}: {t*1e3:8.2f} ms per forward  {t*1e3/B:7.4f} ms per frame  {B/t:8.1f} frames/s", flush=True)
    del det, img; torch.cuda.empty_cache()
