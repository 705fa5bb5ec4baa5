"""Shared by tests/test_sq_survey.py (CPU: oracle, GPU: kernel): how far a fit ends from the reference's own fit,
measured against how far the reference ends from itself when its input moves by one or two float32 ulps.

tests/golden/sq_survey.npz (make_golden_sq_survey.py) holds, for 46 problems of 10 ... 300 views, the reference's
final parameters and fitted box plus the same for ten ulp-level nudges of its initial state.  Per problem:
    e      max relative deviation (floor 1e-3) of OUR final parameters from the reference's
    s      the same for the farthest of the reference's ten nudged runs            (its self-sensitivity)
    eL, sL relative change of the fit loss (sq_libs.py:395-430 evaluated at the final parameters)
    iou    3D IoU of our fitted oriented box with the reference's; siou the smallest IoU among its nudged runs
    fl     how much the loss moves over the last 30 of the 200 steps of one fit, (max - min) / mean: the fits end in a
           jitter state of the L1 loss under Adam's fixed step, not at a minimum (the reference's own per-step records in
           sq_steps.npz show the same 2 % ... 11 %), so this is the resolution below which a loss difference means nothing
"""
import numpy as np

from conftest import rel_err


def survey_rows(z, finals, oracle, dispatch=None):
    """finals [n, 9]: final parameters of the implementation under test for the n survey problems.  dispatch: optional
    tests/golden/sq_survey_dispatch.npz (make_golden_sq_dispatch.py) -- the reference's own un-nudged fits re-run with ATen's
    CPU kernels dispatched to another instruction set; d = its deviation from the stored reference run, the same measure as e"""
    from odam_amd import merge, multi_view
    rows = []
    for i in range(int(z["n_cases"])):
        p0, cls, P, tgt, mask = z[f"c{i}_p0"], int(z[f"c{i}_cls"]), z[f"c{i}_P"], z[f"c{i}_tgt"], z[f"c{i}_mask"]
        ref, nud = z["final"][i], z["nudged_final"][i]
        e = rel_err(finals[i], ref)
        sk = np.array([rel_err(n, ref) for n in nud])
        loss = lambda p: oracle.loss_grad(p, P, tgt, mask, cls, p0[4:7])[0]
        Lr = loss(ref)
        eL = abs(loss(finals[i]) - Lr) / Lr
        sL = max(abs(loss(n) - Lr) / Lr for n in nud)
        tail = oracle.fit(p0, P, tgt, mask, cls, want_loss=True)[2][-30:]
        fl = float((tail.max() - tail.min()) / tail.mean())
        qc = np.asarray(multi_view.compute_oriented_bbox(oracle.points(finals[i]).astype(np.float64)))
        iou = merge.box3d_iou_pairs(qc[None], z["bbox_qc"][i][None])[0][0]
        siou = merge.box3d_iou_pairs(z["nudged_bbox_qc"][i], np.repeat(z["bbox_qc"][i][None], len(nud), 0))[0].min()
        d = {lv: rel_err(dispatch[f"final_{lv}"][i], ref) for lv in ("default", "avx2", "avx512")} if dispatch is not None else {}
        rows.append(dict(i=i, views=int(z["views"][i]), e=e, s=sk.max(), rank=float((sk < e).mean()), eL=eL, sL=sL, fl=fl,
                         iou=float(iou), siou=float(siou), d=d))
    return rows


def check(rows):
    """The rule (task 1a).  Per problem, hard bounds: the deviation from the reference never exceeds three
    times the reference's own ulp-level spread (a single draw compared with the largest of ten), for the parameters,
    for the loss (or the loss's own step-to-step jitter at the end of the fit, whichever is larger) and for the fitted box.  Population: in at least 80 % of the problems the deviation is inside the
    spread itself, wherever the reference is reproducible (spread <= 1e-4) so are we, and our result ranks among the
    reference's own nudged runs like one more of them (mean rank below 0.75; 0.5 = indistinguishable)."""
    for r in rows:
        assert r["e"] <= max(1e-4, 3 * r["s"]), r
        assert r["eL"] <= max(1e-3, 3 * r["sL"], r["fl"]), r
        assert r["iou"] >= min(0.99, 1 - 3 * (1 - r["siou"])), r
        if r["s"] <= 1e-4:
            assert r["e"] <= 1e-4, r
    n = len(rows)
    inside = sum(r["e"] <= max(1e-4, r["s"] * (1 + 1e-3)) for r in rows)
    assert inside >= 0.8 * n, inside
    assert sum(r["eL"] <= max(1e-3, r["sL"] * (1 + 1e-3)) for r in rows) >= 0.8 * n
    assert sum(r["iou"] >= min(0.99, r["siou"] - 1e-3) for r in rows) >= 0.8 * n
    assert np.mean([r["rank"] for r in rows]) <= 0.75
    assert sum(r["e"] <= 1e-4 for r in rows) >= 8          # tracked to the last step on these


def divergence_lines(div):
    """summary lines of tests/golden/sq_divergence.npz (make_sq_divergence.py): what makes the free-running oracle leave the reference's
    trajectory, measured by substituting the reference's own torch operations for the oracle's, part by part"""
    t, e, cause = div["first_param_step"], div["final_dev"], [str(c) for c in div["cause"]]
    n = len(cause)
    has = lambda c: sum(c in x for x in cause)
    return ["first divergence (first step after which the nine parameters differ from the reference's in any bit: step %d ... %d, median %d): "
            "A = the points' transcendentals (torch.cos / sin / exp / log are MKL VML calls in this torch build -- closed source --, abs(x) ** e "
            "is Sleef_powf16_u10 + glibc powf for the scalar tail; the oracle: glibc) change the gradient's bits there in %d of %d problems, "
            "B = the projection (MKL sgemm vs the oracle's k-ordered fma chain) in %d, C = autograd's summation order in %d" % (
                t[:, 0].min(), t[:, 0].max(), int(np.median(t[:, 0])), has("A"), n, has("B"), has("C")),
            "with the reference's OWN transcendentals in the oracle (H1): bit-identical through 200 steps on %d of %d, e <= 1e-4 on %d; with "
            "its projection too (H2): e <= 1e-4 on %d (the oracle alone, H0: %d) -- B is exact already, A and C each suffice to move the last "
            "bits at the first step, and A is not restatable" % (int((t[:, 1] == 0).sum()), n, int((e[:, 1] <= 1e-4).sum()),
                                                                  int((e[:, 2] <= 1e-4).sum()), int((e[:, 0] <= 1e-4).sum()))]


def table(rows, div=None):
    have_d = bool(rows and rows[0].get("d"))
    out = ["# problem views | e (ours vs reference)  s (reference vs its own ulp-nudged runs, max of 10)  rank | loss: eL sL jitter | box IoU: ours, reference's worst nudge"
           + (" | d: the reference re-run on ATen's scalar (default) CPU kernels vs the stored (AVX-512) run" if have_d else "")
           + (" | first divergence: step, cause (A transcendentals: MKL VML / Sleef, B projection: MKL sgemm, C autograd summation order), "
              "e with the reference's own transcendentals" if div is not None else "")]
    for r in rows:
        out.append("%2d %3d | %.2e %.2e %.1f | %.2e %.2e %.2e | %.4f %.4f" % (r["i"], r["views"], r["e"], r["s"], r["rank"], r["eL"], r["sL"],
                                                                           r["fl"], r["iou"], r["siou"])
                   + (" | %.2e" % r["d"]["default"] if have_d else "")
                   + (" | %3d %-3s %.2e" % (div["first_param_step"][r["i"], 0], str(div["cause"][r["i"]]) or "-", div["final_dev"][r["i"], 1])
                      if div is not None else ""))
    n = len(rows)
    out.append("# e <= 1e-4: %d of %d; e <= max(1e-4, s): %d; reference spread s > 1e-4: %d; mean rank %.2f" % (
        sum(r["e"] <= 1e-4 for r in rows), n, sum(r["e"] <= max(1e-4, r["s"] * (1 + 1e-3)) for r in rows),
        sum(r["s"] > 1e-4 for r in rows), np.mean([r["rank"] for r in rows])))
    if have_d:
        out.append("# the reference against itself across ATen CPU dispatch levels (same source, inputs, machine; stored run = AVX-512): "
                   "scalar kernels within 1e-4 on %d of %d (max %.2e), AVX2 identical on %d of %d -- ours within 1e-4 on %d of %d" % (
                       sum(r["d"]["default"] <= 1e-4 for r in rows), n, max(r["d"]["default"] for r in rows),
                       sum(r["d"]["avx2"] == 0 for r in rows), n, sum(r["e"] <= 1e-4 for r in rows), n))
    if div is not None:
        out += ["# " + l for l in divergence_lines(div)]
    return "\n".join(out) + "\n"
