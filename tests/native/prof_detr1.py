import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = sys.argv[2] if len(sys.argv) > 2 else "fp32"
BB = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
det = detector.Detector(backbone=BB, max_batch=B, n_streams=1, dtype=DT); det.load_state_dict(weights.make_state_dict(backbone=BB))
img = torch.randn(B, 3, 800, 1066, device="cuda:0")
for _ in range(2): det(img)
torch.cuda.synchronize(); print("detr forward ok")
