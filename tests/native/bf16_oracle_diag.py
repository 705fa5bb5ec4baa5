"""where the bf16 detector and the bf16-faithful oracle part: layer4 features, encoder memory, heads"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import detr_oracle as O
from odam_amd import detector, weights
bb = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
blocks = (3, 4, 23, 3) if bb == "resnet101" else (3, 4, 6, 3)
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (192, 256)
sd = weights.make_state_dict(backbone=bb, seed=0, scene=True)
det = detector.Detector(backbone=bb, max_batch=2, device="cuda:0", dtype="bf16", n_streams=1); det.load_state_dict(sd)
torch.manual_seed(3); img = torch.randn(2, 3, H, W)
rb = O.detr_forward_bf16(sd, img, blocks=blocks, return_taps=True)
rf = O.detr_forward(sd, img, blocks=blocks, return_taps=True)
out = det(img.cuda())
l4, mem = det.debug_taps(2, H, W)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
def rms(a, b): return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
for name, g, b_, f in (("layer4", l4.cpu(), rb["_layer4"], rf["_layer4"]), ("memory", mem.cpu(), rb["_memory"], rf["_memory"]),
                       ("hs", out["pred_obj_features"].cpu(), rb["pred_obj_features"], rf["pred_obj_features"]),
                       ("logits", out["pred_logits"].cpu(), rb["pred_logits"], rf["pred_logits"])):
    print(f"{name:8s} max-rel: gpu-bf16oracle {rel(g, b_):.2e}  gpu-fp32oracle {rel(g, f):.2e}  bf16oracle-fp32oracle {rel(b_, f):.2e}"
          f" | rms-rel: {rms(g, b_):.2e} {rms(g, f):.2e} {rms(b_, f):.2e}")
