"""The reference's self-sensitivity for the optim_process golden (sq_optim.npz): run_multi_view.optim_process of the
imported reference on the same scene with one world-position column of every track scaled by (1 +- k * 2^-23),
k = 1, 2 -- an ulp-level change of the float32 initial translate -- six runs.  Stores the fitted parameters and boxes
of each run (sq_optim_spread.npz); tests bound the GPU's deviation from the reference by this spread.
Run: python tests/golden/make_golden_optim_spread.py"""
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

NUDGES = [(9, 1), (10, -1), (11, 1), (9, -2), (10, 2), (11, -2)]


def _one(k):
    import torch
    torch.set_num_threads(1)
    import refenv
    refenv.setup()
    import make_golden as G
    from odam_amd import synth
    scene = synth.make_scene(**G.SCENE)
    col, n = NUDGES[k]
    tracks = []
    for t in scene["tracks"]:
        t = t.copy()
        t[:, col] *= 1.0 + n * 2.0 ** -23
        tracks.append(t)
    params, qc, _ = G.ref_optim(dict(scene, tracks=tracks))
    return params, qc


def main():
    with Pool(6) as pool:
        res = pool.map(_one, range(len(NUDGES)))
    np.savez_compressed(os.path.join(HERE, "sq_optim_spread.npz"), nudges=np.asarray(NUDGES, np.int32),
                        params=np.stack([r[0] for r in res]), bboxes_qc=np.stack([r[1] for r in res]))
    z = np.load(os.path.join(HERE, "sq_optim.npz"))
    rel = lambda a, b: np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3), axis=-1)
    print("spread per object:", np.round(rel(np.stack([r[0] for r in res]).astype(np.float64), z["params"][None].astype(np.float64)).max(0), 5))


if __name__ == "__main__":
    main()
