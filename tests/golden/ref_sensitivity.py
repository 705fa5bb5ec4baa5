"""How well conditioned is the reference's own 200-step fit?  Re-runs SuperQuadricOptimizer.run of the imported
reference with ONE initial translate component moved by one float32 ulp and reports the relative change of the
final 9 parameters.  Output of the build container is committed as sq_reference_sensitivity.txt.
Run: python tests/golden/ref_sensitivity.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import refenv; refenv.setup()
import numpy as np
import src.super_quadric.sq_libs as L
from odam_amd import synth
from make_golden import STEP_CASES, params_of
def run(prob, cls, nudge):
    t = np.asarray(prob["translate"], np.float32).copy()
    if nudge: t[nudge - 1] = np.nextafter(t[nudge - 1], np.float32(10))
    opt = L.SuperQuadricOptimizer(t, prob["angle"], prob["dims"], cls, "super_quadric", True)
    return params_of(opt.run(prob["bbox_lines"], None, prob["P"], 200))
rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)))
for ci, (F, seed) in enumerate(STEP_CASES + [(24, 21), (24, 22), (48, 23), (48, 24), (100, 25)]):
    prob = synth.make_sq_problem(F, seed); cls = prob["class_id"]
    base = run(prob, cls, 0)
    print(ci, F, seed, "nudged-vs-base rel:", [round(rel(run(prob, cls, k), base), 6) for k in (1, 2, 3)], flush=True)
