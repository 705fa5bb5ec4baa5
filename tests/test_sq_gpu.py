"""gfx950 SQ-fit kernel (through the C ABI) against the CPU oracle: bit-exact trajectories."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitter():
    from odam_amd import sq
    f = sq.SqFitter("cuda:0", 200)
    yield f
    f.close()


def _pack(probs):
    from odam_amd import sq
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"], p.get("rep", "super_quadric")) for p in probs])
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs])
    tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    vc = [len(p["P"]) for p in probs]
    cls = [p["class_id"] for p in probs]
    return p0, cls, vc, P, tgt, mask


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


@pytest.mark.parametrize("views", [[1, 3, 10, 12], [16, 33, 64, 100], [256, 17], [1024]])
def test_trajectory_bit_exact_vs_oracle(fitter, oracle, views):
    from odam_amd import synth
    probs = [synth.make_sq_problem(F, 100 + F) for F in views]
    p0, cls, vc, P, tgt, mask = _pack(probs)
    out = fitter.fit(p0, cls, vc, P, tgt, mask, n_iters=200, want_loss=True, want_traj=True)
    gp, gt, gl, gpts = (out[k].cpu().numpy() for k in ("params", "traj", "loss", "points"))
    off = 0
    for i, F in enumerate(vc):
        op, otraj, oloss = oracle.fit(p0[i], P[off:off + F], tgt[off:off + F], mask[off:off + F], cls[i],
                                      want_traj=True, want_loss=True)
        assert np.array_equal(_bits(otraj), _bits(gt[i])), f"F={F}"
        assert np.array_equal(_bits(op), _bits(gp[i]))
        assert np.array_equal(_bits(oloss), _bits(gl[i]))
        assert np.array_equal(_bits(oracle.points(op)), _bits(gpts[i]))
        off += F


@pytest.mark.parametrize("rep,prior", [("cube", True), ("quadric", True), ("super_quadric", False)])
def test_representations_and_no_prior(fitter, oracle, rep, prior):
    from odam_amd import synth
    probs = [dict(synth.make_sq_problem(14, 7), rep=rep), dict(synth.make_sq_problem(21, 8), rep=rep)]
    p0, cls, vc, P, tgt, mask = _pack(probs)
    out = fitter.fit(p0, cls, vc, P, tgt, mask, n_iters=60, representation=rep, prior=prior, want_traj=True)
    gt = out["traj"].cpu().numpy()
    off = 0
    for i, F in enumerate(vc):
        _, otraj, _ = oracle.fit(p0[i], P[off:off + F], tgt[off:off + F], mask[off:off + F], cls[i] if prior else -1,
                                 n_iters=60, representation={"super_quadric": 0, "cube": 1, "quadric": 2}[rep],
                                 want_traj=True)
        assert np.array_equal(_bits(otraj), _bits(gt[i]))
        if rep != "super_quadric":
            assert np.array_equal(gt[i][:, 7:], np.repeat(p0[i][None, 7:], 60, 0))  # shapes frozen
        off += F


def test_masked_and_invisible_views(fitter, oracle):
    """all-edges-dropped views, a camera looking away (no valid point: extents stay at the +-1e6 fill)"""
    from odam_amd import synth
    pr = synth.make_sq_problem(12, 3)
    p0, cls, vc, P, tgt, mask = _pack([pr])
    mask[0] = 0; mask[5, 1:] = 0
    P[3] = P[3] * np.float32(-1.0)   # flips the sign of the depth row: every point fails z > 0.5
    out = fitter.fit(p0, cls, vc, P, tgt, mask, n_iters=50, want_traj=True, want_loss=True)
    _, otraj, oloss = oracle.fit(p0[0], P, tgt, mask, cls[0], n_iters=50, want_traj=True, want_loss=True)
    assert np.array_equal(_bits(otraj), _bits(out["traj"].cpu().numpy()[0]))
    assert np.array_equal(_bits(oloss), _bits(out["loss"].cpu().numpy()[0]))


def test_points_batch_bit_exact(fitter, oracle):
    rs = np.random.RandomState(2)
    p = np.concatenate([rs.normal(0, 1, (20, 3)), rs.uniform(-3.5, 3.5, (20, 1)), rs.uniform(0.2, 1.0, (20, 3)),
                        rs.normal(0, 2, (20, 2))], 1).astype(np.float32)
    g = fitter.points(p).cpu().numpy()
    for i in range(len(p)):
        assert np.array_equal(_bits(g[i]), _bits(oracle.points(p[i])))


def test_reference_golden_fits(fitter, oracle, golden):
    """Reference-generated cases: the GPU equals the oracle bit for bit on all of them and follows the reference's own
    trajectory to 1e-5 over the first 40 steps.  Where the free-running fits END relative to the reference is measured
    problem by problem against the reference's own ulp-level spread in tests/test_sq_survey.py (46 problems)."""
    z = golden("sq_steps.npz")
    for c in range(int(z["n_cases"])):
        d = {k[len(f"c{c}_"):]: z[k] for k in z.files if k.startswith(f"c{c}_")}
        cls = int(d["cls"])
        out = fitter.fit(d["p0"][None], [cls], [len(d["tgt"])], d["P"], d["tgt"], d["mask"], want_traj=True)
        gp = out["params"].cpu().numpy()[0]
        op, _, _ = oracle.fit(d["p0"], d["P"], d["tgt"], d["mask"], cls)
        assert np.array_equal(_bits(gp), _bits(op))
        assert rel_err(out["traj"].cpu().numpy()[0][39], d["p_after"][39]) <= 1e-5


def test_reference_golden_fits_at_256_views(fitter, oracle, golden):
    """config 5's size: the reference's own 256-view fits.  GPU == oracle bit for bit; the free-running fit tracks
    the reference to 1e-5 over the first 10 steps (with 1024 (view, direction) winners per step the first ulp-level
    tie that resolves differently comes at step 13 / 18: tests/test_sq_oracle.py::test_first_divergence_is_a_tie).  The
    end state is bounded in tests/test_sq_survey.py (four 256-view problems among the 46)."""
    z = golden("sq_steps_full.npz")
    for c in range(int(z["n_cases"])):
        d = {k[len(f"c{c}_"):]: z[k] for k in z.files if k.startswith(f"c{c}_")}
        cls = int(d["cls"])
        out = fitter.fit(d["p0"][None], [cls], [len(d["tgt"])], d["P"], d["tgt"], d["mask"], want_traj=True)
        op, otraj, _ = oracle.fit(d["p0"], d["P"], d["tgt"], d["mask"], cls, want_traj=True)
        assert np.array_equal(_bits(out["params"].cpu().numpy()[0]), _bits(op))
        assert np.array_equal(_bits(out["traj"].cpu().numpy()[0]), _bits(otraj))
        assert rel_err(otraj[9], d["p_after"][9]) <= 1e-5, c


def test_optim_process_on_gpu_vs_reference(golden):
    from odam_amd.multi_view import optim_process
    z = golden("sq_optim.npz")
    tracks = [z[f"track{i}"] for i in range(int(z["n_tracks"]))]
    out = optim_process(tracks, [int(x) for x in z["img_names"]], z["T_wcs"], z["P_cws"], 480, 640, z["K"],
                        "super_quadric", True, 200, 10, return_params=True)
    assert np.allclose(np.asarray(out["bboxes_dl"]), z["bboxes_dl"], rtol=0, atol=1e-12)
    check_against_spread(out, z["params"], z["bboxes_qc"], golden("sq_optim_spread.npz"))


def check_against_spread(out, ref_params, ref_qc, spread, unfitted_atol=0.0, allow_frac=0.0):
    """fitted objects: deviation from the reference's parameters at most 3x the reference's own spread under small
    perturbations of its input (six or twelve runs stored with the golden), or 1e-4 where it is reproducible; fitted box
    IoU >= 0.99 or no worse than 3x the spread's loss.  `allow_frac`: share of the fitted objects that may leave the
    parameter bound (a handful of sampled perturbations does not exhaust what a last-bit difference can trigger:
    tests/test_sq_survey.py finds 13 % of 46 problems outside the max of ten) -- their boxes must still agree to
    IoU >= 0.97.  Objects that are not fitted (fewer than 10 usable views) keep their initial parameters."""
    from odam_amd import merge
    n = len(ref_params)
    bad, n_fit = [], 0
    for i in range(n):
        if not out["fitted"][i]:      # initial parameters = means over the track's rows (which the caller may hold to 1e-5)
            assert np.abs(np.asarray(out["params"][i], np.float64) - ref_params[i]).max() <= unfitted_atol, i
            continue
        n_fit += 1
        e = rel_err(out["params"][i], ref_params[i])
        s = max(rel_err(p[i], ref_params[i]) for p in spread["params"])
        iou = merge.box3d_iou_pairs(np.asarray(out["bboxes_qc"][i])[None], ref_qc[i][None])[0][0]
        siou = merge.box3d_iou_pairs(spread["bboxes_qc"][:, i], np.repeat(ref_qc[i][None], len(spread["bboxes_qc"]), 0))[0].min()
        if e > max(1e-4, 3 * s):
            bad.append((i, e, s, iou))
            assert iou >= 0.97, (i, e, s, iou)
        else:
            assert iou >= min(0.99, 1 - 3 * (1 - siou)), (i, iou, siou)
    assert len(bad) <= allow_frac * n_fit, bad
    return bad



def test_full_size_batch_properties(fitter, oracle):
    """BASELINE config 5 shape (500 objects x 256 views x 200 steps): determinism, independence of the
    batch order, and spot parity against the oracle."""
    from odam_amd import synth
    n = 500
    base = [synth.make_sq_problem(256, 1000 + s) for s in range(10)]
    rs = np.random.RandomState(0)
    probs = []
    for i in range(n):
        b = dict(base[i % 10])
        b["translate"] = b["translate"] + rs.normal(0, 0.02, 3)
        b["angle"] = b["angle"] + rs.normal(0, 0.05)
        probs.append(b)
    p0, cls, vc, P, tgt, mask = _pack(probs)
    a = fitter.fit(p0, cls, vc, P, tgt, mask, want_points=False)["params"].cpu().numpy()
    b = fitter.fit(p0, cls, vc, P, tgt, mask, want_points=False)["params"].cpu().numpy()
    assert np.array_equal(_bits(a), _bits(b))
    rev = list(range(n))[::-1]
    P3 = P.reshape(n, 256, 12); t3 = tgt.reshape(n, 256, 4); m3 = mask.reshape(n, 256, 4)
    c = fitter.fit(p0[rev], [cls[i] for i in rev], vc, P3[rev].reshape(-1, 12), t3[rev].reshape(-1, 4),
                   m3[rev].reshape(-1, 4), want_points=False)["params"].cpu().numpy()
    assert np.array_equal(_bits(a), _bits(c[::-1]))
    assert np.isfinite(a).all()
    for i in (0, 137, 499):
        op, _, _ = oracle.fit(p0[i], P3[i], t3[i], m3[i], cls[i])
        assert np.array_equal(_bits(op), _bits(a[i]))


def test_limits_are_reported(fitter):
    from odam_amd import synth, _lib
    pr = synth.make_sq_problem(4, 1)
    p0, cls, vc, P, tgt, mask = _pack([pr])
    n = 16 * 1024 + 1        # one view more than 16 workgroups x 1024 rows can hold
    with pytest.raises(_lib.OdamError):
        fitter.fit(p0, cls, [n], np.zeros((n, 12), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32))
    with pytest.raises(KeyError):
        fitter.fit(p0, [9], vc, P, tgt, mask)


def test_more_views_than_one_workgroup_holds(fitter, oracle):
    """3000 and 1500 views per object: beyond the 1024 rows of one workgroup, fitted through the view split --
    trajectory and parameters bit-exact vs the oracle (20 steps keep the serial oracle short)"""
    from odam_amd import synth
    probs = [synth.make_sq_problem(F, 900 + F) for F in (3000, 1500, 40)]
    p0, cls, vc, P, tgt, mask = _pack(probs)
    out = fitter.fit(p0, cls, vc, P, tgt, mask, n_iters=20, want_traj=True)
    gp, gt = out["params"].cpu().numpy(), out["traj"].cpu().numpy()
    off = 0
    for i, F in enumerate(vc):
        op, otraj, _ = oracle.fit(p0[i], P[off:off + F], tgt[off:off + F], mask[off:off + F], cls[i], n_iters=20,
                                  want_traj=True)
        assert np.array_equal(_bits(otraj), _bits(gt[i][:20])), F
        assert np.array_equal(_bits(op), _bits(gp[i])), F
        off += F


def test_view_split_is_bit_identical_on_every_path():
    """The fit of one object split over 2 / 4 / 16 / 32 workgroups -- including the fallback where a workgroup never waits
    for its partners and recomputes their slices (ODAM_SQ_SPLIT_WAIT=0) -- gives the bits of the unsplit kernel
    (parameters, losses and surface points of 14 objects with 10..300 views).  The switches are read once per
    process, hence one subprocess per setting."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(__file__), "native", "sq_split_dump.py")
    digests = {}
    for name, env in (("off", {"ODAM_SQ_SPLIT": "1"}), ("two", {"ODAM_SQ_SPLIT": "2"}), ("four", {"ODAM_SQ_SPLIT": "4"}),
                      ("sixteen", {"ODAM_SQ_SPLIT": "16"}), ("four, no waiting", {"ODAM_SQ_SPLIT": "4", "ODAM_SQ_SPLIT_WAIT": "0"}),
                      ("sixteen, no waiting", {"ODAM_SQ_SPLIT": "16", "ODAM_SQ_SPLIT_WAIT": "0"}),
                      ("thirty-two", {"ODAM_SQ_SPLIT": "32"}), ("thirty-two, no waiting", {"ODAM_SQ_SPLIT": "32", "ODAM_SQ_SPLIT_WAIT": "0"})):
        out = subprocess.run([sys.executable, script], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[name] = [l for l in out.stdout.splitlines() if l.startswith("digest")][0]
    assert len(set(digests.values())) == 1, digests


@pytest.mark.parametrize("kind", ["equal view counts: two 512-thread workgroups per CU", "unequal view counts: longest object first"])
def test_oversubscribed_launch_forms_give_the_same_bits(kind):
    """odam_config sq.two_per_cu on a launch of more objects than CUs (one workgroup per object): objects of about equal view counts run as
    512-thread workgroups, two per CU; objects of unequal view counts are fitted longest first (workgroup b takes object order[b]).  Either way
    every parameter of every object equals the plain form's (1024 threads, object b on workgroup b) bit for bit -- nothing of a fit depends on
    its workgroup's id or size: phases are strided loops, the reductions have fixed shapes.  300 objects, 40 steps.  (The LDS regions that
    share their words since round 6 -- phase B2's queues with the surface points, the dyadic tables with the gradient rows -- are exercised by
    every fit of this file.)"""
    import torch
    from odam_amd import _lib, sq, synth
    equal = kind.startswith("equal")
    probs = [synth.make_sq_problem(24 if equal else 10 + (i * 7) % 61, 900 + i % 40) for i in range(300)]
    p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
    p0[:, 1] += np.linspace(0, 0.03, len(probs)).astype(np.float32)
    tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
    P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs])
    tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
    vc = [len(p["P"]) for p in probs]; cls = [p["class_id"] for p in probs]
    f = sq.SqFitter("cuda:0", 40)
    outs = {}
    old = _lib.get_config("sq.two_per_cu")
    try:
        for mode in (0, 1):
            _lib.set_config("sq.two_per_cu", mode)
            outs[mode] = f.fit(p0, cls, vc, P, tgt, mask, n_iters=40)["params"].cpu().numpy()
    finally:
        _lib.set_config("sq.two_per_cu", old)
    assert np.isfinite(outs[0]).all() and not np.array_equal(outs[0], p0)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
