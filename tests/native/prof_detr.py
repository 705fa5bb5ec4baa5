import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights
det = detector.Detector(max_batch=8); det.load_state_dict(weights.make_state_dict())
img = torch.randn(8, 3, 800, 1066, device="cuda:0")
for _ in range(3): det(img)
torch.cuda.synchronize(); print("detr forward ok")
