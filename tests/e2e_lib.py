"""Shared by tests/test_e2e.py (GPU) and tests/test_e2e_fits_oracle.py (CPU, the oracle standing in for the kernel, which
it equals bit for bit): the fits of the whole-chain fixture on IDENTICAL inputs -- the reference's own tracks -- pinned
object by object against the reference's fit (tests/golden/e2e.npz params<w>, spread_params<w>: the reference re-run under
six 1-2 ulp nudges of its input).

Measured (round 3, both passes; `python -m pytest tests/test_e2e_fits_oracle.py -s` prints the table):
  * 16 of 21 fitted objects of pass 1 and 9 of 12 of pass 2 end within 2.2e-5 of the reference (north-star: 1e-4) --
    including objects whose reference fit moves by 1e-3 ... 1e-1 under its own ulp nudges;
  * INSIDE_REF_SPREAD: objects that end farther than 1e-4 from the un-nudged reference but no farther than the
    reference's own nudged runs do (e <= s: the reference is not reproducible on them at the ulp level);
  * BISTABLE: one object (pass 1 #20 = pass 2 #6) whose fit has exactly two end states 2.74e-4 apart, selected by an
    arg-extreme tie: under 56 ulp-level nudges of the input our fit ends in the reference's state (<= 2e-6) for 36 of them and
    in the other state for 20; un-nudged it is in the other one, and all six stored nudges of the reference happen to stay in
    the first.  The test shows both states exist: un-nudged e <= 3e-4 with box IoU >= 0.999, and the +1 ulp nudge of column 9
    reproduces the reference to 5e-6.
No share of objects is allowed to fall outside: every fitted object is in exactly one of the three classes.

Round 6: the classes are DECIDED AT TEST TIME from the stored spread (no list of object indices): tight if e <= TIGHT; else inside the
reference's spread if e <= s and the box is no worse than the reference's worst nudged box; else the object must be shown bistable -- within
3e-4 / IoU 0.999 un-nudged AND some one-ulp nudge of its input reproduces the reference to 5e-6.  The population bounds (how many
may be in each class) are what round 3 measured.  `fixture_digest` pins the fixture the bounds were measured on."""
import hashlib
import os

import numpy as np

from conftest import GOLDEN, rel_err

SEQ = dict(n=40, h=256, w=320, seed=11)          # make_golden_e2e.py SEQ
TIGHT = 5e-5                                      # measured <= 2.2e-5
MIN_TIGHT = {1: 16, 2: 9}                         # measured: 16 of 21 and 9 of 12 fitted objects
MAX_INSIDE_REF_SPREAD = {1: 4, 2: 2}              # measured: objects 6, 9, 14, 24 of pass 1 and 1, 12 of pass 2
MAX_BISTABLE = {1: 1, 2: 1}                       # measured: pass 1 #20 = pass 2 #6
# sha256 of tests/golden/e2e.npz as committed: the population bounds above were measured on THIS fixture; regenerating it
# (make_golden_e2e.py, or any change to synth.make_sequence) must re-measure them, not inherit them
E2E_SHA256 = "3e57e318d50c674205ad9e57dd94d9c074bb59de3f90603d304bb137433b8b00"
# one-ulp nudges tried on an object that is neither tight nor inside the spread: (column, ulps) of its track rows
BISTABLE_NUDGES = ((9, 1), (9, -1), (10, 1), (10, -1), (11, 1), (11, -1), (12, 1), (12, -1))


def fixture_digest():
    with open(os.path.join(GOLDEN, "e2e.npz"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def check_fixture():
    assert fixture_digest() == E2E_SHA256, ("tests/golden/e2e.npz is not the fixture the class bounds of tests/e2e_lib.py were measured on: "
                                            "re-measure MIN_TIGHT / MAX_* (python -m pytest tests/test_e2e_fits_oracle.py -s) and update E2E_SHA256")


def reference_self_agreement(z):
    """the reference against ITSELF on this fixture: per pass, how many of its fitted objects stay within 1e-4 (the north star's
    tolerance) of the un-nudged run under all six stored 1-2 ulp nudges of its input -- no implementation under test involved"""
    out = {}
    for w in (1, 2):
        ref_p, sp = z[f"params{w}"], z[f"spread_params{w}"]
        fitted = [i for i, t in enumerate(reference_tracks(z, w)) if len(t) >= 10]      # run_multi_view.py:44-46: fewer views are not fitted
        s = np.array([max(rel_err(p[i], ref_p[i]) for p in sp) for i in fitted])
        out[w] = {"objects": len(fitted), "within_1e-4_of_itself": int((s <= 1e-4).sum()), "max": float(s.max())}
    return out


def sequence_geometry():
    from odam_amd import synth
    seq = synth.make_sequence(**SEQ)
    K, T_wcs = seq["K"], np.asarray(seq["T_wcs"])
    P_cws = np.stack([K @ np.linalg.inv(T)[:3, :] for T in T_wcs])
    return seq, K, T_wcs, P_cws


def reference_tracks(z, w):
    name, n = ("track", int(z["n_tracks"])) if w == 1 else ("merged", int(z["n_merged"]))
    return [z[f"{name}{i}"].copy() for i in range(n)]


def check_fits_on_reference_tracks(z, w, fit_tracks, record=None):
    """fit_tracks(list of [n, 82] tracks) -> dict with "params" [n, 9], "bboxes_qc", "fitted" (optim_process, return_params)"""
    from odam_amd import merge
    refs = reference_tracks(z, w)
    out = fit_tracks([t.copy() for t in refs])
    ref_p, ref_qc = z[f"params{w}"], z[f"bboxes_qc{w}"]
    sp, sqc = z[f"spread_params{w}"], z[f"spread_bboxes_qc{w}"]
    check_fixture()
    rows, n_class = [], {"tight": 0, "inside": 0, "bistable": 0}
    for i in range(len(refs)):
        if not out["fitted"][i]:
            assert np.abs(np.asarray(out["params"][i], np.float64) - ref_p[i]).max() <= 1e-6, i      # initial parameters
            continue
        e = rel_err(out["params"][i], ref_p[i])
        s = max(rel_err(p[i], ref_p[i]) for p in sp)
        iou = merge.box3d_iou_pairs(np.asarray(out["bboxes_qc"][i])[None], ref_qc[i][None])[0][0]
        siou = merge.box3d_iou_pairs(sqc[:, i], np.repeat(ref_qc[i][None], len(sqc), 0))[0].min()
        rows.append((w, i, len(refs[i]), e, s, iou, siou))
        if e <= TIGHT and iou >= 0.9999:
            n_class["tight"] += 1
            if record:
                record(f"e2e.fit_on_ref_tracks.tight_pass{w}", e)
        elif e <= s * (1 + 1e-3) and iou >= siou - 1e-4:
            n_class["inside"] += 1      # the reference itself, re-run one ulp away, ends at least this far from its own fit
        else:
            # neither: allowed only for an object with two end states selected by an arg-extreme tie -- close un-nudged, and the
            # reference's own state is reached from an input one ulp away
            assert e <= 3e-4 and iou >= 0.999, (w, i, e, s, iou)
            best = np.inf
            for col, n in BISTABLE_NUDGES:
                t = refs[i].copy()
                t[:, col] *= 1.0 + n * 2.0 ** -23
                best = min(best, rel_err(fit_tracks([t])["params"][0], ref_p[i]))
                if best <= 5e-6:
                    break
            assert best <= 5e-6, (w, i, e, best)
            n_class["bistable"] += 1
            if record:
                record(f"e2e.fit_on_ref_tracks.bistable_nudged_pass{w}", best)
    assert n_class["tight"] >= MIN_TIGHT[w], (w, n_class)
    assert n_class["inside"] <= MAX_INSIDE_REF_SPREAD[w], (w, n_class)
    assert n_class["bistable"] <= MAX_BISTABLE[w], (w, n_class)
    return rows


def table(rows):
    return "\n".join("pass %d obj %2d views %2d: e %.2e  reference's own ulp spread %.2e  box IoU %.4f (reference's worst nudge %.4f)" % r
                     for r in rows)
