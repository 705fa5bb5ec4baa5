import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
for bb, H, W in (("resnet50", 800, 1066), ("resnet101", 800, 1066)):
    sd = weights.make_state_dict(backbone=bb)
    for dtype in ("fp32", "bf16"):
        B = 32
        det = detector.Detector(backbone=bb, max_batch=B, n_streams=1, dtype=dtype); det.load_state_dict(sd)
        img = torch.randn(B, 3, H, W, device="cuda:0")
        for _ in range(2): det(img)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): det(img)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        gf = 163.1 if bb == "resnet50" else 290.0
        print(f"{bb} {dtype} B={B} {H}x{W}: {dt*1e3:.2f} ms/batch = {B/dt:.1f} frames/s = {gf*B/dt/1e3:.1f} TFLOP/s", flush=True)
        det.close(); del det, img; torch.cuda.empty_cache()
