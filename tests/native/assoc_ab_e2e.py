"""persistent matching kernel vs the launch sequence on the frames of the e2e sequence (scene weights, real detector)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
from odam_amd import associator, detector, synth, transforms, weights
from odam_amd.processor import OdamProcess
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scene_weights
DEV = "cuda:0"
SEQ = dict(n=40, h=256, w=320, seed=11)
seq = synth.make_sequence(**SEQ)
det = detector.Detector(max_batch=8, device=DEV, n_streams=1); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
sd = scene_weights.make_scene_associator_state_dict(2, 8, seed=0)
one = associator.Associator(cfg, max_tracks=64, device=DEV); one.load_state_dict(sd); one._handle()
os.environ["ODAM_ASSOC_PERSIST"] = "0"
many = associator.Associator(cfg, max_tracks=64, device=DEV); many.load_state_dict(sd); many._handle()
del os.environ["ODAM_ASSOC_PERSIST"]
orig = one.assignment
log = []
import ctypes
from odam_amd import _lib
def dump(h, which, rows, cols):
    buf = np.zeros((rows, cols), np.float32)
    _lib.check(_lib.lib().odam_assoc_debug_read(h, which, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.size)), "dbg")
    return buf
shown = [0]
def both(tracks, detections, n_tracks, n_det, **kw):
    a = orig(tracks, detections, n_tracks, n_det).cpu().numpy()
    da = [dump(one._handle(), 0, n_tracks + 30, 512), dump(one._handle(), 1, n_tracks + 30, 256), dump(one._handle(), 2, n_tracks, 32)]
    b = many.assignment(tracks, detections, n_tracks, n_det).cpu().numpy()
    db = [dump(many._handle(), 0, n_tracks + 30, 512), dump(many._handle(), 1, n_tracks + 30, 256), dump(many._handle(), 2, n_tracks, 32)]
    if n_tracks == 33 and not shown[0]:
        shown[0] = 1
        for nm, x, y in zip(("X", "mT", "scores"), da, db):
            print(nm, "persist vs launches max diff", np.abs(x - y).max(), "| rows 27 vs 6 persist equal:", np.array_equal(x[27], x[6]), np.abs(x[27]-x[6]).max(),
                  "launches equal:", np.array_equal(y[27], y[6]), np.abs(y[27]-y[6]).max())
        print("scores col 9 rows 27,6,7 persist", da[2][[27, 6, 7], 9], "launches", db[2][[27, 6, 7], 9])
        print("X[27]-X[6] nonzero cols (persist):", np.flatnonzero(da[0][27] != da[0][6])[:20], "mT:", np.flatnonzero(da[1][27] != da[1][6])[:20])
        print("mT persist vs launches row 27 nonzero diff cols", np.flatnonzero(da[1][27] != db[1][27])[:20], "values", da[1][27][:8], db[1][27][:8])
    c = orig(tracks, detections, n_tracks, n_det).cpu().numpy()
    ma = associator.hungarian_matching(torch.from_numpy(np.exp(a[:-1, :-1])), 0.1)
    mb = associator.hungarian_matching(torch.from_numpy(np.exp(b[:-1, :-1])), 0.1)
    d = np.abs(np.exp(a) - np.exp(b))
    log.append((n_tracks, n_det, d.max(), np.unravel_index(d.argmax(), d.shape), np.abs(np.exp(a) - np.exp(c)).max(), np.isfinite(a).all(), np.array_equal(ma, mb)))
    if not np.array_equal(ma, mb) and len(log) < 400:
        np.set_printoptions(precision=3, suppress=True, linewidth=250)
        for c in np.flatnonzero(ma != mb)[:2]:
            pa, pb = np.exp(a[:-1, c]), np.exp(b[:-1, c])
            top = np.argsort(-pa)[:4]
            print(f"T={n_tracks} det {c}: persist -> {ma[c]}, launches -> {mb[c]}; top rows {top} persist {pa[top]} launches {pb[top]} diff {pa[top]-pb[top]}")
            for r in top[:2]:
                print("    row", r, "cols>0.01:", {int(k): (float(np.exp(a[r, k])), float(np.exp(b[r, k]))) for k in np.flatnonzero(np.exp(a[r]) > 0.01)})
    return torch.from_numpy(b).to(DEV)
one.assignment = both
proc = OdamProcess(det, one, transforms.Transforms(size=SEQ["h"]), None)
proc.init_sequence(seq["K"], SEQ["h"], SEQ["w"])
for i, fid in enumerate(seq["img_names"]):
    proc.process_frame(Image.fromarray(seq["frames"][i]), fid, seq["T_wcs"][i])
for l in log: print("T=%d n=%d  |dP| persist-vs-launches %.2e at %s  persist-vs-persist %.2e finite %s same matches %s" % l)
