import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch, numpy as np
from odam_amd import detector, weights, synth
sd = weights.make_state_dict()
frames = torch.randn(512, 3, 800, 1066, device="cuda:0")
for B, ns in [(32, 2), (32, 3), (48, 2), (64, 2), (64, 1)]:
    det = detector.Detector(max_batch=B, n_streams=ns); det.load_state_dict(sd)
    for _ in det.detect_resident_chunks(frames[:2 * B * ns], (640, 480), synth.K_SCANNET, chunk=128): pass
    torch.cuda.synchronize(); t0 = time.time()
    for _ in det.detect_resident_chunks(frames, (640, 480), synth.K_SCANNET, chunk=128): pass
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"B={B} streams={ns}: {512/dt:.1f} frames/s ({163.1*512/dt/1e3:.1f} TFLOP/s)", flush=True)
    det.close(); del det; torch.cuda.empty_cache()
