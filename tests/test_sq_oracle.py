"""The CPU oracle (oracle/sq_oracle.c) against vectors produced by the reference itself
(tests/golden/make_golden.py imported likojack/ODAM in the build container)."""
import ctypes
import os

import numpy as np
import pytest

from conftest import REPO, rel_err


def test_mt19937_draws_are_numpy_randomstate0(oracle):
    d = np.zeros(2000, np.float32)
    oracle.lib.oracle_sq_uniform_draws(0, 2000, oracle.P(d))
    raw = np.random.RandomState(0).randint(0, 2 ** 32, 2000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(d, raw.astype(np.float32) / np.float32(2 ** 32))
    k = (d[1000:] * np.float32(201)).astype(np.int32)  # sampling.cpp:211 never indexes past the buffer
    assert k.min() >= 0 and k.max() <= 200


def test_sampler_bit_exact_vs_reference_binding(oracle, golden):
    z = golden("sq_sampler.npz")
    for a, e, et, om in zip(z["a"], z["e"], z["etas"], z["omegas"]):
        oe, oo = oracle.sample(a, e)
        assert np.array_equal(oe.view(np.uint32), et.view(np.uint32))
        assert np.array_equal(oo.view(np.uint32), om.view(np.uint32))


def test_sampler_bit_exact_vs_reference_cpp_build(oracle):
    """oracle/_ref/libodam_ref_sampler.so is the reference's own sampling.cpp compiled by oracle/Makefile."""
    so = os.path.join(REPO, "oracle", "_ref", "libodam_ref_sampler.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference sources absent on this box)")
    R = ctypes.CDLL(so)
    rs = np.random.RandomState(5)
    fp = oracle.fp
    for _ in range(300):
        a = rs.uniform(0.02, 1.6, 3).astype(np.float32); e = rs.uniform(0.2, 1.6, 2).astype(np.float32)
        et = np.zeros(1000, np.float32); om = np.zeros(1000, np.float32)
        R.sample_on_batch(a.ctypes.data_as(fp), e.ctypes.data_as(fp), et.ctypes.data_as(fp),
                          om.ctypes.data_as(fp), 1, 1, 1000, 201, 0)
        oe, oo = oracle.sample(a, e)
        assert np.array_equal(oe.view(np.uint32), et.view(np.uint32))
        assert np.array_equal(oo.view(np.uint32), om.view(np.uint32))


def _cases(z):
    for c in range(int(z["n_cases"])):
        pre = f"c{c}_"
        yield c, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


@pytest.mark.parametrize("fixture", ["sq_steps.npz", "sq_steps_full.npz"])     # 12..64 views; 256 views (config 5's size)
def test_every_step_matches_reference_when_started_from_its_state(oracle, golden, fixture):
    """Teacher-forced: from the reference's parameters before step t, the oracle's points, loss and
    gradient match the reference's (autograd) and the Adam update lands on the reference's next state."""
    z = golden(fixture)
    for c, d in _cases(z):
        cls = int(d["cls"]); s0 = d["p0"][4:7]
        m = np.zeros(9, np.float32); v = np.zeros(9, np.float32)
        for it in (0, 100, 199):
            pts = oracle.points(d["p"][it])
            assert np.abs(pts - d[f"pts{it}"]).max() <= 4e-7          # a few ulp at |x| ~ 1
        flipped = []
        for it in range(200):
            _, loss, g, _, _ = oracle.loss_grad(d["p"][it], d["P"], d["tgt"], d["mask"], cls, s0)
            rg = d["g"][it]
            assert abs(loss - d["loss"][it]) <= 1e-5 * abs(d["loss"][it])
            # the 9 gradients are sums of up to 4F signed terms of size ~f/z that largely cancel near
            # convergence; float32 summation order (autograd's vs the oracle's fixed tree) shows at 1e-6
            if np.abs(g - rg).max() > max(1e-5 * np.abs(rg).max(), 1e-4):
                flipped.append(it)
            p1, m, v = oracle.adam(d["p"][it], m, v, rg, it + 1)
            assert np.abs(p1 - d["p_after"][it]).max() <= 1.2e-7, (c, it)  # <= 1 ulp
        # The loss is continuous in which surface point wins a view's extreme, the gradient is not: where two of
        # the 1000 points tie to the last ulp of the projected coordinate (the reference evaluates them with torch's
        # vector math library, the oracle with libm) the winner, and with it one of the 4F terms, differs.  Never
        # at 12..64 views; 2 and 10 of 200 steps in the two 256-view cases.
        assert len(flipped) <= (0 if len(d["tgt"]) <= 64 else 20), (c, flipped)


def test_free_running_trajectory_tracks_reference(oracle, golden):
    """Free-running 200-step fits.  The reference's own trajectory is a chain of discrete decisions
    (arg-min/max winners, sign(ext - tgt), sampler rounding); the oracle follows it to ~1e-6 until
    the first decision that differs, after which both settle in different jitter states of the L1
    objective.  Asserted: tight tracking over the first 40 steps for every case, and the north-star
    tolerance (1e-4 relative) on the final parameters for the cases whose decision chain survives."""
    z = golden("sq_steps.npz")
    finals = {}
    for c, d in _cases(z):
        out, traj, _ = oracle.fit(d["p0"], d["P"], d["tgt"], d["mask"], int(d["cls"]), want_traj=True)
        ref = d["p_after"]
        assert np.array_equal(ref[-1], d["final_plain"])  # stepwise re-drive == SuperQuadricOptimizer.run
        assert rel_err(traj[39], ref[39]) <= 1e-5, c
        finals[c] = rel_err(out, d["final_plain"])
    stable = [c for c, e in finals.items() if e <= 1e-4]
    assert {1, 2} <= set(stable), finals


@pytest.mark.parametrize("fixture", ["sq_steps.npz", "sq_steps_full.npz"])
def test_first_divergence_is_a_tie(oracle, golden, fixture):
    """Why free-running fits leave the reference's trajectory: every step takes 4F discrete
    decisions per direction -- which of the 1000 surface points is a view's extreme, and on which side of the target edge
    the extreme lies (the sign of the L1 term).  The oracle follows the reference to ~1e-6 until one of those decisions
    comes out differently, and this test shows that the FIRST such decision is a tie at the resolution of the state
    difference accumulated so far: the two candidate points project within 2e-4 px of each other (a few float32 ulps of
    a coordinate of some hundred pixels; measured 4e-6 ... 4e-5 px), or the extreme sits that close to the target edge.  The reference's decisions at step t are read off the oracle evaluated at the
    reference's own state (valid wherever its gradient reproduces the reference's autograd gradient, which the
    teacher-forced test above asserts); the oracle's own decisions are those at its free-running state."""
    z = golden(fixture)
    n_div = 0
    for c, d in _cases(z):
        cls = int(d["cls"]); s0 = d["p0"][4:7]
        P = d["P"].reshape(-1, 3, 4).astype(np.float64); tgt = d["tgt"]; mask = d["mask"] > 0
        _, traj, _ = oracle.fit(d["p0"], d["P"], d["tgt"], d["mask"], cls, want_traj=True)
        dev = [rel_err(traj[t], d["p_after"][t]) for t in range(200)]
        first_dev = next((t for t in range(200) if dev[t] > 1e-4), None)
        first_dec = None
        for t in range(200 if first_dev is None else first_dev + 1):
            own = d["p0"] if t == 0 else traj[t - 1]
            _, _, g_r, ext_r, win_r = oracle.loss_grad(d["p"][t], d["P"], tgt, d["mask"], cls, s0)
            _, _, _, ext_o, win_o = oracle.loss_grad(own, d["P"], tgt, d["mask"], cls, s0)
            sign_r, sign_o = np.sign(ext_r - tgt), np.sign(ext_o - tgt)
            diff_w = (win_r != win_o) & mask
            diff_s = (sign_r != sign_o) & mask
            if not (diff_w.any() or diff_s.any()):
                continue
            # candidate winners that are different POINTS (the 1000 samples contain exact duplicates) and project apart
            pts = oracle.points(d["p"][t]).astype(np.float64)
            uvw = np.einsum("fij,nj->fni", P, np.c_[pts, np.ones(len(pts))])
            uv = uvw[..., :2] / (np.abs(uvw[..., 2:]) + 1e-6)
            gaps = []
            for f, k in zip(*np.nonzero(diff_w)):
                gaps.append(abs(uv[f, win_r[f, k], k // 2] - uv[f, win_o[f, k], k // 2]))
            edge = [abs(float(ext_r[f, k]) - float(tgt[f, k])) for f, k in zip(*np.nonzero(diff_s))]
            real = [x for x in gaps if x > 0] + edge
            if not real:
                continue            # only duplicates of one point swapped places: same gradient
            first_dec = t
            assert max(real) <= 2e-4, (c, t, gaps, edge)     # pixels; non-tied candidates are 0.1 ... 10 px apart
            break
        if first_dev is not None:
            n_div += 1
            # the trajectories agree until a decision differs, never the other way round
            assert first_dec is not None and first_dec <= first_dev, (c, first_dec, first_dev)
            assert max(dev[:first_dec]) <= 1e-4 if first_dec else True
    assert n_div >= (2 if "full" in fixture else 3)      # the fixtures do contain diverging cases


@pytest.mark.parametrize("case", [0, 3, 5, 9])
def test_autograd_restatement_is_the_reference_fit(golden, case):
    """oracle/sq_autograd_oracle.py restates the reference's fit the way the reference computes it -- torch CPU tensors, autograd,
    torch.optim.Adam, anomaly mode on (sq_libs.py:432-475) -- so that bench.py's cpu_baseline can time the reference's ALGORITHM on a
    box without /root/reference.  Same ATen kernels in the same order: the nine parameters after 200 steps equal the reference's own
    run of the problem (tests/golden/sq_survey.npz, `final`) bit for bit, with anomaly detection on and off."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import sq_autograd_oracle as ag
    z = golden("sq_survey.npz")
    icov = np.load(os.path.join(REPO, "odam_amd", "data", "scale_prior.npz"))["icov"].astype(np.float32).reshape(8, 3, 3)
    keep = torch.get_num_threads()
    torch.set_num_threads(1)      # the fixture's setting (make_golden_sq_survey.py:41)
    try:
        p0, cls = z[f"c{case}_p0"], int(z[f"c{case}_cls"])
        out = ag.fit(p0, z[f"c{case}_P"], z[f"c{case}_tgt"], z[f"c{case}_mask"], icov[cls], anomaly=(case != 3))
    finally:
        torch.set_num_threads(keep)
    assert np.array_equal(out.view(np.uint32), z["final"][case].view(np.uint32)), (out, z["final"][case])
