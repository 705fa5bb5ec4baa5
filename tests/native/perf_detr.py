import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
sd = weights.make_state_dict()
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32").split(",")]:
    det = detector.Detector(max_batch=B); det.load_state_dict(sd)
    img = torch.randn(B, 3, 800, 1066, device="cuda:0")
    for _ in range(2): det(img)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n): det(img)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"B={B}: {dt*1e3:.2f} ms/batch = {B/dt:.1f} frames/s = {163.1*B/dt/1e3:.1f} TFLOP/s", flush=True)
    det.close(); del det, img; torch.cuda.empty_cache()
