"""Golden vectors of the reference Associator (src/models/associator.py) on CPU, deterministic synthetic weights
(odam_amd.weights.make_associator_state_dict) and synthetic track / detection tensors in the layout
OdamProcess builds (processor.py:49-68, 155-179).  Run: python tests/golden/make_golden_assoc.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG = {"descriptor_dim": 256, "weights": "indoor", "keypoint_encoder": [78, 256, 256],
       "GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "match_threshold": 0.2,
       "sinkhorn_iterations": 100}


def make_inputs(T, n_det, seed):
    rs = np.random.RandomState(seed)
    tracks = -np.ones((T, 79, 100), np.float32)
    centres = rs.uniform(-1, 1, (T, 3)) + np.array([0, 0, 2.5])
    for t in range(T):
        n = int(rs.randint(1, 101))
        f0 = int(rs.randint(0, 400))
        tracks[t, 0, :n] = f0 + np.arange(n) * 10
        tracks[t, 1, :n] = rs.randint(0, 8)
        bb = rs.uniform(0.1, 0.6, 2)
        tracks[t, 2:6, :n] = np.r_[bb, bb + rs.uniform(0.1, 0.3, 2)][:, None] + rs.normal(0, 0.01, (4, n))
        tracks[t, 6:9, :n] = rs.uniform(0.3, 1.5, 3)[:, None] * rs.uniform(0.95, 1.05, (3, n))
        tracks[t, 9:12, :n] = centres[t][:, None] + rs.normal(0, 0.05, (3, n))
        a = rs.uniform(-np.pi, np.pi)
        tracks[t, 12, :n] = np.sin(a); tracks[t, 13, :n] = np.cos(a)
        tracks[t, 14, :n] = rs.uniform(0.7, 1.0, n)
    det = -np.ones((1, 79, 30), np.float32)
    pick = rs.permutation(max(T, n_det))[:n_det] % T
    det[0, 0, :n_det] = 5000
    det[0, 1, :n_det] = rs.randint(0, 8, n_det)
    bb = rs.uniform(0.1, 0.6, (2, n_det))
    det[0, 2:4, :n_det] = bb; det[0, 4:6, :n_det] = bb + rs.uniform(0.1, 0.3, (2, n_det))
    det[0, 6:9, :n_det] = rs.uniform(0.3, 1.5, (3, n_det))
    det[0, 9:12, :n_det] = centres[pick].T + rs.normal(0, 0.08, (3, n_det))
    a = rs.uniform(-np.pi, np.pi, n_det)
    det[0, 12, :n_det] = np.sin(a); det[0, 13, :n_det] = np.cos(a)
    det[0, 14, :n_det] = rs.uniform(0.7, 1.0, n_det)
    return tracks, det


CASES = [(1, 1, 0), (4, 3, 1), (17, 30, 2), (40, 12, 3)]


def main():
    import refenv
    refenv.setup()
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)
    from src.models.associator import Associator
    from odam_amd import weights
    model = Associator(CFG)
    model.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0), strict=True)
    model.eval()
    data = {"n_cases": np.int32(len(CASES))}
    for ci, (T, n_det, seed) in enumerate(CASES):
        tr, de = make_inputs(T, n_det, seed)
        with torch.no_grad():
            out = model({"tracks": torch.from_numpy(tr), "detections": torch.from_numpy(de), "valid_list": [(T, n_det)]},
                        0.1, eval_only=True, device="cpu")
        data[f"c{ci}_tracks"] = tr; data[f"c{ci}_det"] = de
        data[f"c{ci}_Z"] = out["pred"][0][0].numpy(); data[f"c{ci}_matches"] = np.asarray(out["matches"][0])
        print("case", ci, "T", T, "n_det", n_det, "matches", out["matches"][0][:8])
    torch.Tensor.to = orig_to
    # Sinkhorn alone on a random score matrix
    from src.models.associator import log_optimal_transport
    torch.manual_seed(5)
    sc = torch.randn(1, 23, 17) * 2
    data["ot_scores"] = sc[0].numpy()
    data["ot_Z"] = log_optimal_transport(sc, torch.tensor(1.0), 100)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "assoc.npz"), **data)


if __name__ == "__main__":
    main()
