"""diagnostics of tests/test_e2e.py: deviation of GPU detections / tracks / fits from the reference run"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from PIL import Image
from odam_amd import associator, detector, synth, transforms, weights, merge
from odam_amd.processor import OdamProcess
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scene_weights
from conftest import rel_err
z = np.load(os.path.join(ROOT, "tests/golden/e2e.npz"))
seq = synth.make_sequence(n=40, h=256, w=320, seed=11)
det = detector.Detector(max_batch=8, device="cuda:0", n_streams=1); det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}, max_tracks=64)
net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))
proc = OdamProcess(det, net, transforms.Transforms(size=256), None); proc.init_sequence(seq["K"], 256, 320)
rows = {}; run = proc.run_detector
def logged(rgb, fid, T):
    rows[fid] = run(rgb, fid, T); return rows[fid]
proc.run_detector = logged
for i, fid in enumerate(seq["img_names"]):
    proc.process_frame(Image.fromarray(seq["frames"][i]), fid, seq["T_wcs"][i])
dd = [np.abs(np.asarray(rows[f], np.float64).reshape(-1, 79) - z[f"det{f}"]).max() for f in seq["img_names"] if np.asarray(rows[f]).reshape(-1,79).shape == z[f"det{f}"].shape]
print("detections: frames with identical shape", len(dd), "max abs diff %.2e median %.2e" % (max(dd), np.median(dd)))
print("tracks", len(proc.tracks), "ref", int(z["n_tracks"]))
td = [np.abs(t - z[f"track{i}"]).max() for i, t in enumerate(proc.tracks) if t.shape == z[f"track{i}"].shape]
print("tracks identical shape", len(td), "max abs diff %.2e median %.2e" % (max(td), np.median(td)))
out1 = proc.optim_process([t.copy() for t in proc.tracks])
P1 = np.stack([q.params for q in out1["quadrics"]])
outr = proc.optim_process([z[f"track{i}"].copy() for i in range(int(z["n_tracks"]))])      # our fit on the reference's tracks
Pr = np.stack([q.params for q in outr["quadrics"]])
for i in range(len(P1)):
    qc, dl = np.asarray(out1["bboxes_qc"][i]), np.asarray(out1["bboxes_dl"][i])
    if not np.array_equal(qc, dl):
        s = max(rel_err(p[i], z["params1"][i]) for p in z["spread_params1"])
        iou = merge.box3d_iou_pairs(qc[None], z["bboxes_qc1"][i][None])[0][0]
        print("obj %2d len %2d: e(ours on our tracks) %.2e  e(ours on ref tracks) %.2e  ref ulp-spread %.2e  iou %.4f" % (i, len(proc.tracks[i]), rel_err(P1[i], z["params1"][i]), rel_err(Pr[i], z["params1"][i]), s, iou))
