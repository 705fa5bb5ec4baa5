"""Golden for the plugin surface: the reference's OdamProcess (src/processor.py) is driven over a synthetic
sequence with canned per-frame detections (its run_detector is replaced by a table lookup) and a small
deterministic stand-in associator; the resulting `tracks`, `T_wcs`, `P_cws`, `usable_frames` are stored.
odam_amd.processor.OdamProcess driven the same way must reproduce them (tests/test_processor.py).
Run: python tests/golden/make_golden_process.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def canned_detections(n_frames=40, seed=4):
    """per frame: list of 79-float rows in run_detector's layout (processor.py:276-288)"""
    from odam_amd import synth
    sc = synth.make_scene(n_frames, 5, seed=seed, min_views=12, max_views=n_frames)
    rs = np.random.RandomState(seed)
    per = {fid: [] for fid in sc["img_names"]}
    for t in sc["tracks"]:
        for row in t:
            fid = int(row[0])
            i = sc["img_names"].index(fid)
            T_cw = np.linalg.inv(sc["T_wcs"][i])
            t_co = (T_cw @ np.r_[row[9:12], 1.0])[:3]
            ang = rs.uniform(-np.pi, np.pi)
            per[fid].append([float(fid), float(row[1]), row[2] / 640, row[3] / 480, row[4] / 640, row[5] / 480,
                             row[6], row[7], row[8], t_co[0], t_co[1], t_co[2], np.sin(ang), np.cos(ang),
                             float(rs.uniform(0.7, 1.0))] + [-1.0] * 64)
    return sc, per


class FakeAssociator:
    """deterministic stand-in with the reference associator's call signature and output dict
    (src/models/associator.py:202-268): score = exp(-|t_co(det) - t_co(last track step)|), greedy arg-max match"""

    def __call__(self, data, threshold, eval_only=True):
        tr = data["tracks"].cpu()            # [T, 79, 100]
        det = data["detections"].cpu()       # [1, 79, 30]
        nt, nd = data["valid_list"][0]
        last = []
        for t in range(nt):
            valid = (tr[t, 0] != -1).nonzero()[:, 0]
            last.append(tr[t, 9:12, valid[-1]])
        last = torch.stack(last)             # [T, 3]
        d = det[0, 9:12, :nd].T              # [nd, 3]
        dist = torch.cdist(last.double(), d.double()).float()
        score = torch.exp(-dist)             # [T, nd]
        pred = torch.full((1, nt + 1, nd + 1), -20.0)
        pred[0, :nt, :nd] = torch.log(score.clamp_min(1e-9))
        matches = np.full(nd, -1.0)
        for j in range(nd):
            i = int(score[:, j].argmax())
            if score[i, j] > 0.7:
                matches[j] = i
        return {"pred": [pred], "loss": 0, "matches": [matches]}


def main():
    import refenv
    refenv.setup()
    from src.processor import OdamProcess
    sc, per = canned_detections()
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)
    proc = OdamProcess(None, FakeAssociator(), None, None)
    proc.init_sequence(sc["K"], 480, 640)
    proc.run_detector = lambda rgb, fid, T: [list(r) for r in per[fid]]
    for i, fid in enumerate(sc["img_names"]):
        proc.process_frame(None, fid, sc["T_wcs"][i])
    torch.Tensor.to = orig_to
    d = dict(n_tracks=np.int32(len(proc.tracks)), usable_frames=np.asarray(proc.usable_frames), T_wcs=np.asarray(proc.T_wcs),
             P_cws=np.asarray(proc.P_cws), K=sc["K"], img_names=np.asarray(sc["img_names"]), scene_T_wcs=sc["T_wcs"])
    for i, t in enumerate(proc.tracks):
        d[f"track{i}"] = t
    for fid, rows in per.items():
        d[f"det{fid}"] = np.asarray(rows, np.float64).reshape(-1, 79)
    np.savez_compressed(os.path.join(HERE, "process_tracks.npz"), **d)
    print("process golden:", len(proc.tracks), "tracks, lengths", [len(t) for t in proc.tracks])

    # the same sequence with the reference's REAL Associator (CPU, synthetic weights): pins the association network
    # inside the plugin surface (tests/test_assoc_gpu.py::test_process_sequence_with_network)
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)
    from src.models.associator import Associator
    from make_golden_assoc import CFG
    from odam_amd import weights
    net = Associator(CFG)
    net.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0), strict=True)
    net.eval()
    fwd = net.forward
    # With these random weights the assignment matrix is nearly uniform: the Hungarian step and the thresholds decide on
    # differences of 1e-5, below what two correct fp32 evaluations of the network agree to (round 2: one detection moved
    # between two GPU code paths that agree to 3e-5 in every entry).  So the fixture also keeps every frame's assignment
    # matrix: the GPU test compares its own matrix with the reference's frame by frame and lets the REFERENCE's matrix take
    # the decision, which keeps the tracks of later frames identical to the reference's whatever the rounding.
    zs = []

    def forward(data, thr, eval_only=False):
        out = fwd(data, thr, eval_only=eval_only, device="cpu")
        zs.append(out["pred"][0][0].detach().numpy().astype(np.float32))
        return out
    net.forward = forward
    proc = OdamProcess(None, net, None, None, score_threshold=0.2)
    proc.init_sequence(sc["K"], 480, 640)
    proc.run_detector = lambda rgb, fid, T: [list(r) for r in per[fid]]
    for i, fid in enumerate(sc["img_names"]):
        proc.process_frame(None, fid, sc["T_wcs"][i])
    torch.Tensor.to = orig_to
    d2 = dict(n_tracks=np.int32(len(proc.tracks)), n_calls=np.int32(len(zs)))
    for i, zz in enumerate(zs):
        d2[f"Z{i}"] = zz
    for i, t in enumerate(proc.tracks):
        d2[f"track{i}"] = t
    np.savez_compressed(os.path.join(HERE, "process_tracks_net.npz"), **d2)
    print("process golden with the reference associator:", len(proc.tracks), "tracks, lengths", [len(t) for t in proc.tracks])


if __name__ == "__main__":
    main()
