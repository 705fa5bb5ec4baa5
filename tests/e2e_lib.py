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
No share of objects is allowed to fall outside: every fitted object is in exactly one of the three classes."""
import numpy as np

from conftest import rel_err

SEQ = dict(n=40, h=256, w=320, seed=11)          # make_golden_e2e.py SEQ
TIGHT = 5e-5                                      # measured <= 2.2e-5
INSIDE_REF_SPREAD = {1: (6, 9, 14, 24), 2: (1, 12)}
BISTABLE = {1: (20,), 2: (6,)}


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
    rows, n_tight = [], 0
    for i in range(len(refs)):
        if not out["fitted"][i]:
            assert np.abs(np.asarray(out["params"][i], np.float64) - ref_p[i]).max() <= 1e-6, i      # initial parameters
            continue
        e = rel_err(out["params"][i], ref_p[i])
        s = max(rel_err(p[i], ref_p[i]) for p in sp)
        iou = merge.box3d_iou_pairs(np.asarray(out["bboxes_qc"][i])[None], ref_qc[i][None])[0][0]
        siou = merge.box3d_iou_pairs(sqc[:, i], np.repeat(ref_qc[i][None], len(sqc), 0))[0].min()
        rows.append((w, i, len(refs[i]), e, s, iou, siou))
        if i in BISTABLE[w]:
            assert e <= 3e-4 and iou >= 0.999, (w, i, e, iou)
            t = refs[i].copy()
            t[:, 9] *= 1.0 + 2.0 ** -23
            other = fit_tracks([t])
            e2 = rel_err(other["params"][0], ref_p[i])
            assert e2 <= 5e-6, (w, i, e2)                   # the reference's end state, one ulp away
            if record:
                record(f"e2e.fit_on_ref_tracks.bistable_nudged_pass{w}", e2)
        elif i in INSIDE_REF_SPREAD[w]:
            assert e <= s * (1 + 1e-3) and iou >= siou - 1e-4, (w, i, e, s, iou, siou)
        else:
            assert e <= TIGHT and iou >= 0.9999, (w, i, e, s, iou)
            n_tight += 1
            if record:
                record(f"e2e.fit_on_ref_tracks.tight_pass{w}", e)
    assert n_tight >= (16 if w == 1 else 9), (w, n_tight)
    return rows


def table(rows):
    return "\n".join("pass %d obj %2d views %2d: e %.2e  reference's own ulp spread %.2e  box IoU %.4f (reference's worst nudge %.4f)" % r
                     for r in rows)
