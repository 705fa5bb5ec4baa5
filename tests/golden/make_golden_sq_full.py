"""Per-step records of the reference's SuperQuadricOptimizer at BASELINE config 5's size: 256 views per object
(likojack/ODAM src/super_quadric/sq_libs.py:432-475 re-driven step by step as in make_golden.py::step_records,
2 problems x 200 steps).  Writes tests/golden/sq_steps_full.npz.   Run: python tests/golden/make_golden_sq_full.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

CASES = [(256, 21), (256, 22)]


def main():
    import refenv
    refenv.setup()
    import make_golden as G
    from odam_amd import synth, sq
    data = {}
    for ci, (F, seed) in enumerate(CASES):
        prob = synth.make_sq_problem(F, seed)
        cls = prob["class_id"]
        rec, pts = G.step_records(prob, cls)
        tgt, mask = sq.lines_to_targets(prob["bbox_lines"])
        pre = f"c{ci}_"
        data[pre + "p0"] = sq.init_params(prob["translate"], prob["angle"], prob["dims"])
        data[pre + "cls"] = np.int32(cls)
        data[pre + "P"] = prob["P"].astype(np.float32).reshape(-1, 12)
        data[pre + "tgt"] = tgt; data[pre + "mask"] = mask
        for k, v in rec.items():
            data[pre + k] = v
        for it, v in pts.items():
            data[pre + f"pts{it}"] = v
        data[pre + "final_plain"] = G.run_plain(prob, cls)
    data["n_cases"] = np.int32(len(CASES))
    np.savez_compressed(os.path.join(HERE, "sq_steps_full.npz"), **data)
    print("sq_steps_full.npz written:", {k: data[k].shape for k in ("c0_P", "c0_p", "c0_g")})


if __name__ == "__main__":
    main()
