"""persistent matching kernel vs the launch sequence on every frame of the 40-frame golden sequence: max |dP| per frame"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import associator, weights, sq
from odam_amd.processor import OdamProcess
DEV = "cuda:0"
z = np.load(os.path.join(ROOT, "tests/golden/process_tracks.npz"), allow_pickle=True)
cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
sd = weights.make_associator_state_dict(2, 8, seed=0)
one = associator.Associator(cfg, max_tracks=64, device=DEV); one.load_state_dict(sd); one._handle()
os.environ["ODAM_ASSOC_PERSIST"] = "0"
many = associator.Associator(cfg, max_tracks=64, device=DEV); many.load_state_dict(sd); many._handle()
del os.environ["ODAM_ASSOC_PERSIST"]
orig = one.assignment
log = []
def both(tracks, detections, n_tracks, n_det, **kw):
    Za = orig(tracks, detections, n_tracks, n_det)
    Zb = many.assignment(tracks, detections, n_tracks, n_det)
    Zc = orig(tracks, detections, n_tracks, n_det)
    a, b, c = Za.cpu().numpy(), Zb.cpu().numpy(), Zc.cpu().numpy()
    log.append((n_tracks, n_det, np.abs(np.exp(a) - np.exp(b)).max(), np.abs(np.exp(a) - np.exp(c)).max(), np.isfinite(a).all()))
    return Zb
one.assignment = both
class _Det: device = DEV
proc = OdamProcess(_Det(), one, None, None, score_threshold=0.2, fitter=sq.SqFitter(DEV, 1))
proc.init_sequence(z["K"], 480, 640)
for i, fid in enumerate(z["img_names"]):
    proc.process_frame(None, int(fid), z["scene_T_wcs"][i], detections=[list(r) for r in z[f"det{int(fid)}"]])
for l in log: print("T=%d n=%d  |dP| persist-vs-launches %.2e  persist-vs-persist %.2e finite %s" % l)
