"""How far may a correct implementation end from the reference's fit?  A survey over 46 problems.

For each synthetic multi-view problem (10 ... 300 views, four of them at BASELINE config 5's 256 views) the reference's
own SuperQuadricOptimizer.run (likojack/ODAM src/super_quadric/sq_libs.py:432-475, imported here) is run
  * once as it is                                      -> final 9 parameters, oriented box of the fitted surface
  * ten more times with its INITIAL state moved by a few float32 ulps (one translate component by +-1 ulp, all three
    by +-2 ulps, the yaw by +-1 ulp)                   -> the same outputs: the reference's self-sensitivity,
i.e. how much of the final state is decided by last-bit effects of the reference's own arithmetic.
Stored: inputs (p0, class, P, tgt, mask) and those outputs.  tests/test_sq_survey.py compares the GPU fit's deviation
from the reference with this spread, problem by problem.
Run: python tests/golden/make_golden_sq_survey.py        (about 5 minutes on 7 processes)
"""
import os
import sys
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

CASES = ([(F, 300 + i) for i, F in enumerate((10, 10, 11, 12, 12, 14, 16, 16, 18, 20, 20, 24, 24, 28, 32, 32, 36, 40, 40, 48,
                                              48, 56, 64, 64, 72, 80, 80, 96, 100, 112, 128, 128, 144, 160, 176, 200, 200, 224,
                                              256, 256, 256, 256, 280, 300, 300, 300))])
# (which, ulps): which = 0..2 a translate component, 3 = all three translate components, 4 = yaw
NUDGES = [(0, 1), (0, -1), (1, 1), (1, -1), (2, 1), (2, -1), (3, 2), (3, -2), (4, 1), (4, -1)]


def _ulps(x, n):
    x = np.float32(x)
    for _ in range(abs(n)):
        x = np.nextafter(x, np.float32(np.inf if n > 0 else -np.inf))
    return x


def _one(job):
    ci, F, seed, k = job
    import torch
    torch.set_num_threads(1)
    import refenv
    refenv.setup()
    import src.super_quadric.sq_libs as L
    import src.utils.box_utils as bu
    from make_golden import params_of
    from odam_amd import synth
    prob = synth.make_sq_problem(F, seed)
    t = np.asarray(prob["translate"], np.float32).copy()
    ang = np.float32(prob["angle"])
    if k >= 0:
        which, n = NUDGES[k]
        if which < 3:
            t[which] = _ulps(t[which], n)
        elif which == 3:
            t = np.array([_ulps(v, n) for v in t], np.float32)
        else:
            ang = _ulps(ang, n)
    opt = L.SuperQuadricOptimizer(t, ang, prob["dims"], prob["class_id"], "super_quadric", True)
    Q = opt.run(prob["bbox_lines"], None, prob["P"], 200)
    pts, _ = Q.compute_ellipsoid_points(use_numpy=True)
    qc = bu.compute_oriented_bbox(pts)          # run_multi_view.py:66-67
    return ci, k, params_of(Q), np.asarray(qc, np.float64)


def main():
    from odam_amd import synth, sq
    jobs = [(ci, F, seed, k) for ci, (F, seed) in enumerate(CASES) for k in range(-1, len(NUDGES))]
    jobs.sort(key=lambda j: -j[1])
    with Pool(7) as pool:
        res = pool.map(_one, jobs, chunksize=1)
    n = len(CASES)
    final = np.zeros((n, 9), np.float32); qc = np.zeros((n, 8, 3))
    nfinal = np.zeros((n, len(NUDGES), 9), np.float32); nqc = np.zeros((n, len(NUDGES), 8, 3))
    for ci, k, p, b in res:
        if k < 0:
            final[ci], qc[ci] = p, b
        else:
            nfinal[ci, k], nqc[ci, k] = p, b
    data = dict(n_cases=np.int32(n), views=np.asarray([c[0] for c in CASES], np.int32), seeds=np.asarray([c[1] for c in CASES], np.int32),
                final=final, bbox_qc=qc, nudged_final=nfinal, nudged_bbox_qc=nqc, nudges=np.asarray(NUDGES, np.int32))
    for ci, (F, seed) in enumerate(CASES):
        prob = synth.make_sq_problem(F, seed)
        tgt, mask = sq.lines_to_targets(prob["bbox_lines"])
        data[f"c{ci}_p0"] = sq.init_params(prob["translate"], prob["angle"], prob["dims"])
        data[f"c{ci}_cls"] = np.int32(prob["class_id"])
        data[f"c{ci}_P"] = prob["P"].astype(np.float32).reshape(-1, 12)
        data[f"c{ci}_tgt"] = tgt; data[f"c{ci}_mask"] = mask
    np.savez_compressed(os.path.join(HERE, "sq_survey.npz"), **data)
    rel = lambda a, b: np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3), axis=-1)
    s = rel(nfinal.astype(np.float64), final[:, None].astype(np.float64)).max(1)
    print("sq_survey.npz:", n, "problems; reference self-sensitivity > 1e-4 on", int((s > 1e-4).sum()))
    for ci in range(n):
        print(ci, CASES[ci], "%.2e" % s[ci])


if __name__ == "__main__":
    main()
