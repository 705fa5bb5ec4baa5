"""gfx950 association network (GNN + fused Sinkhorn) against the reference Associator's outputs
(tests/golden/assoc.npz, made by importing src/models/associator.py on the CPU)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sinkhorn_kernel_vs_reference(golden):
    from odam_amd import _lib
    z = golden("assoc.npz")
    sc = torch.from_numpy(z["ot_scores"]).to(DEV).contiguous()
    m, n = sc.shape
    out = torch.empty(m + 1, n + 1, device=DEV)
    _lib.check(_lib.lib().odam_assoc_sinkhorn(_lib.ptr(sc), n, m, n, ctypes.c_float(1.0), 100, _lib.ptr(out),
                                              ctypes.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)),
               "sinkhorn")
    assert np.abs(out.cpu().numpy() - z["ot_Z"]).max() <= 2e-4     # 100 log-space iterations in fp32


def _sinkhorn_f64(sc, alpha, iters):
    """log_optimal_transport (associator.py:283-312) in float64"""
    m, n = sc.shape
    Z = np.full((m + 1, n + 1), alpha, np.float64); Z[:m, :n] = sc
    norm = -np.log(m + n)
    log_mu = np.full(m + 1, norm); log_mu[-1] = np.log(n) + norm
    log_nu = np.full(n + 1, norm); log_nu[-1] = np.log(m) + norm
    u, v = np.zeros(m + 1), np.zeros(n + 1)
    lse = lambda x, ax: np.log(np.exp(x - x.max(ax, keepdims=True)).sum(ax)) + x.max(ax)
    for _ in range(iters):
        u = log_mu - lse(Z + v[None, :], 1)
        v = log_nu - lse(Z + u[:, None], 0)
    return Z + u[:, None] + v[None, :] - norm


@pytest.mark.parametrize("m,n,scale", [(1, 1, 3.0), (5, 7, 3.0), (42, 21, 8.0), (63, 30, 8.0), (64, 31, 4.0), (100, 12, 8.0), (127, 30, 20.0),
                                       (128, 5, 3.0), (300, 30, 8.0), (42, 21, 150.0), (90, 30, 400.0),
                                       (5, 4, 1.0), (5, 3, 2.0), (40, 20, 0.5), (20, 30, 300.0)])
def test_sinkhorn_rows_on_lanes_vs_float64(m, n, scale):
    """The one-wavefront Sinkhorn kernel (odam_config assoc.sk_fast; <= 128 rows incl. the dustbin) on every rung of its ladder -- scaling
    factors absorbed every 10 iterations, after every iteration, log space -- and the sixteen-wave log-space kernel (more rows, or
    assoc.sk_fast = 0) against the float64 iteration: scores of every size, incl. the +-1000 the hand-built scene weights reach and a
    constant offset of 100 against the dustbin score (process_tracks_net.npz is such a case)."""
    from odam_amd import _lib
    rng = np.random.default_rng(m * 1000 + n)
    sc = (rng.standard_normal((m, n)) * scale).astype(np.float32)
    if (m + n) % 3 == 0:
        sc += 100.0
    ref = _sinkhorn_f64(sc.astype(np.float64), 1.0, 100)
    d = torch.from_numpy(sc).to(DEV)
    outs = []
    for fast in (1, 2, 3, 0):
        _lib.set_config("assoc.sk_fast", fast)
        try:
            out = torch.empty(m + 1, n + 1, device=DEV)
            _lib.check(_lib.lib().odam_assoc_sinkhorn(_lib.ptr(d), n, m, n, ctypes.c_float(1.0), 100, _lib.ptr(out),
                                                      ctypes.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)), "sinkhorn")
            outs.append(out.cpu().numpy().astype(np.float64))
        finally:
            _lib.set_config("assoc.sk_fast", 1)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))       # float32 sums of <= 129 terms, absolute in log space
    for o in outs:
        assert np.isfinite(o).all()
        assert np.abs(o - ref).max() <= tol, (np.abs(o - ref).max(), tol)
        assert (np.abs(np.exp(o) - np.exp(ref)) / np.maximum(np.exp(ref), 1.0)).max() <= 2 * tol      # entries are probabilities x (m + n)


@pytest.mark.parametrize("merge", [1, 0], ids=["merge projection folded into the MLP", "merge projection as its own layer"])
def test_forward_vs_reference_associator(golden, merge):
    """the association forward against the reference's own (src/models/associator.py:202-254 run on the CPU, tests/golden/assoc.npz), in both
    forms of odam_config assoc.merge: the attention's merge projection as its own layer (what the reference computes, step by step) and folded
    into the first MLP layer at load time (the default: one launch / one barrier stage per layer fewer) -- the same bounds for both"""
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    a = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                               "sinkhorn_iterations": 100}, max_tracks=64, device=DEV)
    a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    old = _lib.get_config("assoc.merge")
    _lib.set_config("assoc.merge", merge)
    try:
        a._handle()                                        # the switch is read when the handle is finalised
    finally:
        _lib.set_config("assoc.merge", old)
    for c in range(int(z["n_cases"])):
        tr, de, Zr, mr = z[f"c{c}_tracks"], z[f"c{c}_det"], z[f"c{c}_Z"], z[f"c{c}_matches"]
        T, n_det = Zr.shape[0] - 1, Zr.shape[1] - 1
        out = a({"tracks": torch.from_numpy(tr), "detections": torch.from_numpy(de), "valid_list": [(T, n_det)]}, 0.1,
                eval_only=True)
        Z = out["pred"][0][0].cpu().numpy()
        assert Z.shape == Zr.shape
        # log-assignment: compare where the reference assigns non-negligible mass, and the probabilities everywhere
        assert np.abs(np.exp(Z) - np.exp(Zr)).max() <= 2e-4, c
        big = Zr > -6
        assert np.abs(Z[big] - Zr[big]).max() <= 2e-3, c
        assert np.array_equal(out["matches"][0], mr), c          # object IDs: exact
    a.close()


def test_folded_merge_projection_is_the_same_network(golden):
    """odam_config assoc.merge = 1 against 0 on few and many tracks, on all three forms of the matching layers (rows dealt to the XCDs, device-
    wide barriers, launch sequence): W0' = [W0x | W0m Wm] is the same function of the weights -- assignment probabilities equal to fp32 rounding
    noise (the folded product is rounded once where the layer-by-layer form rounds the message and re-multiplies it)"""
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    sd = weights.make_associator_state_dict(2, 8, seed=0)
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    nets = {}
    old = (_lib.get_config("assoc.merge"), _lib.get_config("assoc.persist"))
    try:
        for merge in (0, 1):
            for persist in (2, 1, 0):
                _lib.set_config("assoc.merge", merge); _lib.set_config("assoc.persist", persist)
                n_ = associator.Associator(cfg, max_tracks=160, device=DEV)
                n_.load_state_dict(sd)
                n_._handle()
                nets[(merge, persist)] = n_
    finally:
        _lib.set_config("assoc.merge", old[0]); _lib.set_config("assoc.persist", old[1])
    g = torch.Generator().manual_seed(9)
    base_t = torch.from_numpy(z["c0_tracks"]); base_d = torch.from_numpy(z["c0_det"])
    worst = 0.0
    for T, n in ((1, 1), (5, 30), (17, 9), (42, 21), (70, 30), (129, 17)):
        tr = base_t[torch.randint(0, base_t.shape[0], (T,), generator=g)].clone()
        tr[:, 1:] += 0.05 * torch.randn(tr[:, 1:].shape, generator=g) * (tr[:, 1:] != -1)
        Z = {k: v.assignment(tr.to(DEV), base_d.to(DEV), T, n).cpu().numpy() for k, v in nets.items()}
        for persist in (2, 1, 0):
            a_, b_ = Z[(0, persist)], Z[(1, persist)]
            assert np.isfinite(b_).all()
            worst = max(worst, float(np.abs(np.exp(a_) - np.exp(b_)).max()))
        assert np.array_equal(Z[(1, 2)], Z[(1, 1)])        # the two persistent kernels stay bit-identical in the folded form too
    assert worst <= 5e-5, worst
    for n_ in nets.values():
        n_.close()


def test_persistent_matching_kernel_vs_launch_sequence(golden):
    """the one-launch matching GNN (grid barriers between the stages, Sinkhorn in workgroup 0) against the same layers as
    ~50 separate launches (config assoc.persist = 0): same assignment matrix to fp32 reorder noise, for few and many tracks"""
    from odam_amd import associator, weights
    z = golden("assoc.npz")
    sd = weights.make_associator_state_dict(2, 8, seed=0)
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    one = associator.Associator(cfg, max_tracks=320, device=DEV)
    one.load_state_dict(sd)
    from odam_amd import _lib
    _lib.set_config("assoc.persist", 0)
    try:
        many = associator.Associator(cfg, max_tracks=320, device=DEV)
        many.load_state_dict(sd)
        many._handle()                                     # the switch is read when the handle is finalised
    finally:
        _lib.set_config("assoc.persist", 2)
    g = torch.Generator().manual_seed(5)
    base_t = torch.from_numpy(z["c0_tracks"]); base_d = torch.from_numpy(z["c0_det"])
    for T, n in ((1, 1), (3, 30), (17, 9), (40, 21), (70, 30), (96, 5), (300, 30), (129, 17)):    # > 64 tracks: several key chunks
        tr = base_t[torch.randint(0, base_t.shape[0], (T,), generator=g)].clone()
        tr[:, 1:] += 0.05 * torch.randn(tr[:, 1:].shape, generator=g) * (tr[:, 1:] != -1)
        de = base_d.clone()
        Za = one.assignment(tr.to(DEV), de.to(DEV), T, n).cpu().numpy()
        Zb = many.assignment(tr.to(DEV), de.to(DEV), T, n).cpu().numpy()
        assert Za.shape == Zb.shape == (T + 1, n + 1)
        assert np.isfinite(Za).all()
        assert np.abs(np.exp(Za) - np.exp(Zb)).max() < 1e-4, (T, n, np.abs(np.exp(Za) - np.exp(Zb)).max())   # measured <= 4e-5
        sig = Zb > -6
        assert np.abs(Za[sig] - Zb[sig]).max() < 2e-3, (T, n)
    # repeated launches keep the barrier counter consistent
    for _ in range(50):
        Zc = one.assignment(tr.to(DEV), de.to(DEV), T, n)
    assert np.array_equal(Zc.cpu().numpy(), Za)
    one.close(); many.close()


def test_lost_persistent_launch_is_rerun_and_not_sticky(golden):
    """A persistent matching launch that cannot get through a device-wide barrier (forced: barrier bound 0) is abandoned,
    reported through the handle's lost-launch counter with Z = NaN, and the Associator re-runs that frame through the launch
    sequence in the same process; the next frame takes the persistent path again and works (the flag is per launch)."""
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    a = associator.Associator(cfg, max_tracks=64, device=DEV)
    a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    L, h = _lib.lib(), a._handle()
    on, cap, need = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.odam_assoc_info(h, ctypes.byref(on), ctypes.byref(cap), ctypes.byref(need)), "info")
    assert need.value == 256 and cap.value >= need.value and on.value == 1, (on.value, cap.value, need.value)   # a whole MI355X holds the grid
    tr, de, Zr = z["c0_tracks"], z["c0_det"], z["c0_Z"]
    T, n = Zr.shape[0] - 1, Zr.shape[1] - 1
    data = {"tracks": torch.from_numpy(tr), "detections": torch.from_numpy(de), "valid_list": [(T, n)]}
    good = a(data, 0.1, eval_only=True)
    seq = a.assignment(torch.from_numpy(tr), torch.from_numpy(de), T, n, sequence=True).cpu().numpy()
    assert a.lost_launches == 0
    # the raw entry point: NaN out and the counter moves
    _lib.check(L.odam_assoc_set_barrier_timeout_us(h, ctypes.c_longlong(0)), "timeout")
    Zraw = a.assignment(torch.from_numpy(tr), torch.from_numpy(de), T, n).cpu().numpy()
    cnt = ctypes.c_uint()
    _lib.check(L.odam_assoc_lost_launches(h, ctypes.byref(cnt)), "lost")
    assert np.isnan(Zraw).all() and cnt.value == 1
    a._lost_seen = cnt.value
    # the Associator call: re-run through the launch sequence, same matches, counted
    out = a(data, 0.1, eval_only=True)
    assert a.lost_launches == 1
    assert np.array_equal(out["pred"][0][0].cpu().numpy(), seq)
    assert np.array_equal(out["matches"][0], good["matches"][0])
    # not sticky: with the normal bound the persistent path works again, bit for bit what it gave before
    _lib.check(L.odam_assoc_set_barrier_timeout_us(h, ctypes.c_longlong(20000)), "timeout")
    again = a(data, 0.1, eval_only=True)
    assert a.lost_launches == 1
    assert np.array_equal(again["pred"][0][0].cpu().numpy(), good["pred"][0][0].cpu().numpy())
    # the switch of the handle: launch sequence on request
    _lib.check(L.odam_assoc_set_persistent(h, 0), "set_persistent")
    off = a(data, 0.1, eval_only=True)
    assert np.array_equal(off["pred"][0][0].cpu().numpy(), seq)
    a.close()


def test_row_partitioned_kernel_checks_its_placement(golden):
    """ADVICE (round 4): gnn_rowpart_kernel exchanges its row-wise stages inside one XCD and is right only if the 32 workgroups of a
    group run behind one L2.  Every launch now compares the hardware XCC ids inside each group: (1) on this box the check passes on
    every launch (no lost launch over 40 frames: the mapping the kernel was built on holds here); (2) a forced mismatch abandons the
    launch, the frame is re-run through the launch sequence with the same result, and after three in a row the handle stays on the
    launch sequence."""
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    a = associator.Associator(cfg, max_tracks=64, device=DEV)
    a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    assert _lib.config()["assoc.persist"] == 2
    L, h = _lib.lib(), a._handle()
    tr, de, Zr = z["c0_tracks"], z["c0_det"], z["c0_Z"]
    T, n = Zr.shape[0] - 1, Zr.shape[1] - 1
    data = {"tracks": torch.from_numpy(tr), "detections": torch.from_numpy(de), "valid_list": [(T, n)]}
    good = a(data, 0.1, eval_only=True)
    for _ in range(40):
        again = a(data, 0.1, eval_only=True)
    assert a.lost_launches == 0, "a group of the row-partitioned kernel straddles XCDs on this box"
    assert np.array_equal(again["pred"][0][0].cpu().numpy(), good["pred"][0][0].cpu().numpy())
    seq = a.assignment(torch.from_numpy(tr), torch.from_numpy(de), T, n, sequence=True).cpu().numpy()
    _lib.check(L.odam_assoc_debug_misplace(h, 1), "misplace")
    on = ctypes.c_int()
    for k in range(3):
        out = a(data, 0.1, eval_only=True)
        assert a.lost_launches == k + 1
        assert np.array_equal(out["pred"][0][0].cpu().numpy(), seq)
        assert np.array_equal(out["matches"][0], good["matches"][0])
    _lib.check(L.odam_assoc_info(h, ctypes.byref(on), None, None), "info")
    assert on.value == 0 and getattr(a, "persistent_given_up", False)      # three in a row: the launch sequence from here on
    out = a(data, 0.1, eval_only=True)
    assert a.lost_launches == 3 and np.array_equal(out["pred"][0][0].cpu().numpy(), seq)
    a.close()


def test_workspace_growth_respects_native_limit(golden):
    """growth past max_tracks builds a larger handle (capped at the native 1024); more tracks than that is a clear error and
    leaves the old handle usable"""
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    cfg = {"GNN_layers": ["self", "cross"], "self_GNN_layers": ["self"], "sinkhorn_iterations": 10}
    a = associator.Associator(cfg, max_tracks=300, device=DEV)
    a.load_state_dict(weights.make_associator_state_dict(1, 2, seed=0))
    base_t = torch.from_numpy(z["c0_tracks"]); de = torch.from_numpy(z["c0_det"])
    tr = base_t[torch.arange(601) % base_t.shape[0]].clone()
    Z = a.assignment(tr, de, 601, 5)
    assert a.max_tracks == 601 and np.isfinite(Z.cpu().numpy()).all()
    tr2 = base_t[torch.arange(700) % base_t.shape[0]].clone()
    a.assignment(tr2, de, 700, 5)
    assert a.max_tracks == 1024
    with pytest.raises(_lib.OdamError, match="native limit"):
        a.assignment(tr2, de, 1025, 5)
    assert a.max_tracks == 1024
    assert np.isfinite(a.assignment(tr, de, 601, 5).cpu().numpy()).all()
    a.close()


def test_process_sequence_with_network(golden):
    """The whole per-frame loop (canned detections -> _prepare_tracks -> association network -> Hungarian ->
    track attachment) on the GPU against the reference's OdamProcess + reference Associator run on the CPU.
    With this fixture's random weights the assignment matrix is nearly uniform and the Hungarian step / thresholds decide on
    differences of ~1e-5 -- less than two correct fp32 evaluations of the network agree to -- so the comparison is frame by
    frame: our assignment matrix must equal the reference's (process_tracks_net.npz, Z<i>) to 2e-4 in every probability,
    the REFERENCE's matrix then takes the decision, and the tracks that result must be the reference's, row for row.
    Frames whose decision does not hang on such a tie must give the same matches from our matrix as well."""
    from odam_amd import associator, sq, weights
    from odam_amd.processor import OdamProcess
    z = golden("process_tracks.npz")
    ref = golden("process_tracks_net.npz")
    net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                                 "sinkhorn_iterations": 100}, max_tracks=64, device=DEV)
    net.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    plain = net.assignment
    calls, same_decision, decisive = [], [], []

    def forced(tracks, detections, n_tracks, n_det, **kw):
        i = len(calls)
        Zr = ref[f"Z{i}"]
        Z = plain(tracks, detections, n_tracks, n_det).cpu().numpy()
        assert Z.shape == Zr.shape, i
        P, Pr = np.exp(Z), np.exp(Zr)
        calls.append(np.abs(P - Pr).max())
        ours = associator.hungarian_matching(torch.from_numpy(P[:-1, :-1]), 0.1)
        theirs = associator.hungarian_matching(torch.from_numpy(Pr[:-1, :-1]), 0.1)
        same_decision.append(np.array_equal(ours, theirs))
        # decisive: no entry within 1e-3 of a threshold and no two candidates of a detection / a track within 1e-3
        core = Pr[:-1, :-1]
        m = min(np.abs(core - 0.1).min(), np.abs(core - 0.2).min())
        for mat in (core, core.T):
            if mat.shape[0] > 1:
                srt = np.sort(mat, axis=0)
                m = min(m, (srt[-1] - srt[-2]).min())
        decisive.append(m > 1e-3)
        return torch.from_numpy(Zr).to(DEV)
    net.assignment = forced

    class _Det:
        device = DEV
    proc = OdamProcess(_Det(), net, None, None, score_threshold=0.2, fitter=sq.SqFitter(DEV, 1))
    proc.init_sequence(z["K"], 480, 640)
    for i, fid in enumerate(z["img_names"]):
        proc.process_frame(None, int(fid), z["scene_T_wcs"][i], detections=[list(r) for r in z[f"det{int(fid)}"]])
    assert len(calls) == int(ref["n_calls"])
    assert max(calls) <= 2e-4, max(calls)
    assert all(s for s, d in zip(same_decision, decisive) if d)
    assert len(proc.tracks) == int(ref["n_tracks"])
    for i, t in enumerate(proc.tracks):
        r = ref[f"track{i}"]
        assert t.shape == r.shape, i
        assert np.array_equal(t[:, :2], r[:, :2])
        assert np.allclose(t, r, rtol=1e-6, atol=1e-6)
    net.close()


def test_process_frames_fast_path_equals_the_frame_by_frame_path(golden):
    """OdamProcess.process_frames (one native call per frame -- odam_assoc_step --, the host's work under the device's, the track list
    extended lazily) against the loop with every step a separate call (src/processor.py:306-345 as restated in _process_frame_host), on
    the canned-detection sequence with the scene association weights: every assignment matrix and every track row equal BIT FOR BIT;
    also frame by frame through process_frame, with the track list read -- and once edited -- in between (the mirror is re-checked after
    every exposure), and with an abandoned persistent launch in the middle (re-run through the launch sequence)."""
    import os, sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import scene_weights
    from odam_amd import _lib, associator, sq
    from odam_amd.processor import OdamProcess
    z = golden("process_tracks.npz")
    ids = [int(f) for f in z["img_names"]]
    T = [z["scene_T_wcs"][i] for i in range(len(ids))]
    dets = [np.asarray(z[f"det{f}"], np.float64).reshape(-1, 79) for f in ids]
    assert len(ids) >= 30 and sum(len(d) for d in dets) >= 60

    class _Det:
        device = DEV

    def make():
        net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                                     "sinkhorn_iterations": 100}, max_tracks=64, device=DEV)
        net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))
        proc = OdamProcess(_Det(), net, None, None, fitter=sq.SqFitter(DEV, 1))
        proc.init_sequence(z["K"], 480, 640)
        return proc, net

    def log_Z(net, Zs):
        fin = net.step_finish

        def wrapped():
            Z = fin()
            Zs.append(Z.clone().numpy())
            return Z
        net.step_finish = wrapped

    # A: every step a separate call; the assignment matrices logged through the associator's call
    proc_a, net_a = make()
    Za = []
    for i in range(len(ids)):
        proc_a._process_frame_host(None, ids[i], T[i], dets[i])
    tracks_a = [t.copy() for t in proc_a.tracks]
    assert len(tracks_a) >= 2 and max(len(t) for t in tracks_a) >= 10
    # ... once more with the matrices kept (assignment_checked returns them)
    proc_a2, net_a2 = make()
    chk = net_a2.assignment_checked

    def logged(*a, **k):
        Zt = chk(*a, **k)
        Za.append(Zt.cpu().numpy().copy())
        return Zt
    net_a2.assignment_checked = logged
    for i in range(len(ids)):
        proc_a2._process_frame_host(None, ids[i], T[i], dets[i])

    def same_tracks(got):
        assert len(got) == len(tracks_a)
        for t, r in zip(got, tracks_a):
            assert t.shape == r.shape and np.array_equal(t.view(np.uint8), np.ascontiguousarray(r).view(np.uint8))

    # B: the whole sequence in one process_frames call
    proc_b, net_b = make()
    Zb = []
    log_Z(net_b, Zb)
    assert proc_b._fast_ok()
    proc_b.process_frames(ids, T, dets)
    assert len(Zb) == len(Za) >= 20
    for a_, b_ in zip(Za, Zb):
        assert a_.shape == b_.shape and np.array_equal(a_.view(np.uint32), b_.view(np.uint32))
    same_tracks(proc_b.tracks)
    assert proc_b.usable_frames == proc_a.usable_frames and len(proc_b.P_cws) == len(ids)
    assert all(np.array_equal(p, q) for p, q in zip(proc_b.P_cws, proc_a.P_cws))
    # the mirror's host-side bookkeeping was brought in step when the list was read
    assert proc_b._win.in_step(proc_b.tracks)

    # C: frame by frame through process_frame (the reference's call), chunks of 3 through process_frames in between; the list is read
    # every fifth frame (after every exposure the mirror is checked against it again)
    proc_c, net_c = make()
    i = 0
    while i < len(ids):
        if i % 7 == 3:
            proc_c.process_frames(ids[i:i + 3], T[i:i + 3], dets[i:i + 3]); i += 3
        else:
            proc_c.process_frame(None, ids[i], T[i], detections=dets[i]); i += 1
        if i % 5 == 0:
            assert sum(len(t) for t in proc_c.tracks) > 0
    same_tracks(proc_c.tracks)
    # ... and an EDIT of the list between two frames is noticed (equal lengths, another last observation): one more frame on both paths
    k = max(range(len(tracks_a)), key=lambda j: len(tracks_a[j]))
    for pr in (proc_a, proc_c):
        pr.tracks[k][-1, 9:12] += 0.25
    extra = dets[-1].copy(); extra[:, 0] = ids[-1] + 1
    proc_a._process_frame_host(None, ids[-1] + 1, T[-1], extra)
    proc_c.process_frame(None, ids[-1] + 1, T[-1], detections=extra)
    ta, tc = proc_a.tracks, proc_c.tracks
    assert len(ta) == len(tc) and sum(len(t) for t in ta) > sum(len(t) for t in tracks_a)
    for t, r in zip(tc, ta):
        assert t.shape == r.shape and np.array_equal(np.ascontiguousarray(t).view(np.uint8), np.ascontiguousarray(r).view(np.uint8))

    # D: an abandoned persistent launch in the middle (every barrier wait of frame 12 times out): re-run through the sequence, same bits
    proc_d, net_d = make()
    launch = net_d.step_launch
    n_launch = [0]

    def lossy(*a, **k):
        n_launch[0] += 1
        if n_launch[0] == 12:
            _lib.check(_lib.lib().odam_assoc_set_barrier_timeout_us(net_d._handle(), 0), "timeout")
        launch(*a, **k)
        if n_launch[0] == 12:
            _lib.check(_lib.lib().odam_assoc_set_barrier_timeout_us(net_d._handle(), 20000), "timeout")
    net_d.step_launch = lossy
    info = (ctypes.c_int(), ctypes.c_int(), ctypes.c_int())
    _lib.check(_lib.lib().odam_assoc_info(net_d._handle(), ctypes.byref(info[0]), ctypes.byref(info[1]), ctypes.byref(info[2])), "info")
    proc_d.process_frames(ids, T, dets)
    if info[0].value:      # the persistent path is in use on this device
        assert net_d.lost_launches == 1
    got = proc_d.tracks
    assert len(got) == len(tracks_a)
    for t, r in zip(got, tracks_a):      # the launch sequence sums in another order than the persistent kernel: same decisions, rows equal
        assert t.shape == r.shape and np.array_equal(t.view(np.uint8), np.ascontiguousarray(r).view(np.uint8))
    # E: a native call of the fast path fails (here: before it reaches the library, with the previous frame's attach already taken off the
    # pending list): the exception comes out, the mirror's content is unknown -- the next frame, on either path, rebuilds it from the track
    # list; retried from the failed frame the sequence ends where the others do
    proc_e, net_e = make()
    launch_e = net_e.step_launch
    n_e = [0]

    def failing(*a, **k):
        n_e[0] += 1
        if n_e[0] == 9:
            raise RuntimeError("launch failed on purpose")
        return launch_e(*a, **k)
    net_e.step_launch = failing
    with pytest.raises(RuntimeError, match="on purpose"):
        proc_e.process_frames(ids, T, dets)
    done = len(proc_e.usable_frames) - 1          # the failed frame's bookkeeping is in; its association never ran
    proc_e._rewind_frame()
    assert 5 <= done < len(ids) - 5
    assert not proc_e._win.in_step(proc_e.tracks)      # nobody may trust the mirror now
    proc_e.process_frames(ids[done:], T[done:], dets[done:])
    same_tracks(proc_e.tracks)
    # F: the corners of process_frame's contract (src/processor.py:312-323): a frame without detections in the middle (bookkeeping only), one
    # with more than 30 (truncated), an empty FIRST frame (tracks start at the second) -- both paths, same inputs
    dets_f = [d.copy() for d in dets]
    dets_f[0] = np.zeros((0, 79))
    dets_f[6] = np.zeros((0, 79))
    many = np.concatenate([dets[9]] * 12)[:34].copy()
    many[:, 9:12] += np.linspace(0.0, 0.5, len(many))[:, None]
    dets_f[9] = many
    proc_h, net_h = make()
    for i in range(len(ids)):
        proc_h._process_frame_host(None, ids[i], T[i], dets_f[i])
    proc_f, net_f = make()
    proc_f.process_frames(ids, T, dets_f)
    th, tf = proc_h.tracks, proc_f.tracks
    assert len(th) == len(tf) >= 2 and proc_h.usable_frames == proc_f.usable_frames == ids
    for t, r in zip(tf, th):
        assert t.shape == r.shape and np.array_equal(np.ascontiguousarray(t).view(np.uint8), np.ascontiguousarray(r).view(np.uint8))
    for n_ in (net_a, net_a2, net_b, net_c, net_d, net_e, net_h, net_f):
        n_.close()


def _random_tracks(rs, lengths):
    tracks = []
    for n in lengths:
        t = rs.normal(0, 1, (n, 82)); t[:, 14:78] = -1
        t[:, 0] = np.arange(n) + rs.randint(0, 50)         # frame ids ascend inside a track
        tracks.append(t)
    return tracks


@pytest.mark.gpu
def test_track_windows_rebuild_and_queued_appends_equal_the_host_path():
    """The device mirror of the track list (odam_trackwin_*) against OdamProcess._preprocess_tracks on the host:
    (a) a bulk rebuild (odam_trackwin_load) from 40 tracks of 1..260 observations (ring wrap past the window of 100, more than
    32 tracks); (b) the append path, 300 back-to-back appends queued behind a long kernel on the same stream -- the pinned
    staging ring has eight slots and every upload from it is asynchronous, so a slot must not be rewritten before its copy has
    run; (c) a track list edited at equal lengths is noticed (marks) and rebuilt."""
    import torch
    from odam_amd.associator import TrackWindows
    from odam_amd.processor import OdamProcess, get_cam_azi
    rs = np.random.RandomState(11)
    proc = OdamProcess(None, None, None, None)
    proc.init_sequence(np.array([[577.87, 0, 319.5], [0, 577.87, 239.5], [0, 0, 1.0]]), 480, 640)
    Q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
    T_wc = np.eye(4); T_wc[:3, :3] = Q; T_wc[:3, 3] = rs.normal(size=3)
    cam_azi = get_cam_azi(T_wc)
    lengths = [1, 2, 99, 100, 101, 260, 37, 7, 8, 9, 128, 129, 136, 257] + list(rs.randint(1, 230, 26))
    tracks = _random_tracks(rs, lengths)
    proj = rs.uniform(-200, 900, (len(tracks), 4))

    def host(tr):
        return np.ascontiguousarray(proc._preprocess_tracks(tr, T_wc, cam_azi, proj_px=proj).transpose(0, 2, 1))

    def device(win):
        out = win.build(torch.from_numpy(proj).to("cuda:0"), np.linalg.inv(T_wc), cam_azi, 640, 480)
        return out.cpu().numpy()

    def host_params(tr):
        """the means a track's surface is evaluated from, as the reference's _prepare_tracks takes them (src/processor.py:188-193:
        np.mean of the azimuth, centre and dimension columns, np.clip, sqrt(dims / 2), shapes -0) -> float32 rows (sq.init_params)"""
        p = np.zeros((len(tr), 9)); p[:, 7:] = -0.0
        for i, t in enumerate(tr):
            p[i, :3] = np.mean(t[:, 9:12], axis=0)
            p[i, 3] = np.mean(t[:, 12], axis=0)
            p[i, 4:7] = np.sqrt(np.clip(np.mean(t[:, 6:9], axis=0), a_min=0.05, a_max=np.inf) / 2)
        return p.astype(np.float32)

    def same_bits(a, b):
        return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))

    win = TrackWindows("cuda:0")
    win.load(tracks)                                                    # (a)
    assert win.in_step(tracks)
    assert np.array_equal(device(win), host(tracks))
    assert same_bits(win.params(), host_params(tracks))                 # running sums of the store == numpy's reductions, bit for bit

    # (b) observation j of every track that has one, one append call each (ids unique per call), all queued behind ~50 ms of
    # device work so that far more than eight uploads are pending when the host refills the ring
    win.reset()
    busy = torch.randn(4096, 4096, device="cuda:0")
    for _ in range(40):
        busy = busy @ busy * 1e-3
    for j in range(max(lengths)):
        ids = [i for i, t in enumerate(tracks) if len(t) > j]
        win.append(ids, np.stack([tracks[i][j] for i in ids]))
    assert win.in_step(tracks)
    assert np.array_equal(device(win), host(tracks))
    assert same_bits(win.params(), host_params(tracks))                 # ... extended append by append (pairwise azimuth sums of 1..260 values)
    del busy

    # (c) same lengths, one track replaced: the marks differ, OdamProcess._windows() would rebuild
    edited = [t.copy() for t in tracks]
    edited[5] = _random_tracks(rs, [len(tracks[5])])[0]
    assert not win.in_step(edited)
    win.load(edited)
    assert np.array_equal(device(win), host(edited))
    assert same_bits(win.params(), host_params(edited))
    # (d) the whole chain in one native call (odam_trackwin_build_tracks: sums -> parameter rows -> surface extents -> window tensor)
    # against the same steps taken one by one
    from odam_amd import sq
    fitter = sq.SqFitter("cuda:0", 1)
    K = np.array([[577.87, 0, 319.5], [0, 577.87, 239.5], [0, 0, 1.0]])
    T_cw = np.linalg.inv(T_wc)
    one = fitter.build_track_windows(win, T_cw, K, cam_azi, 640, 480).cpu().numpy()
    proj_dev = fitter.project_extents(host_params(edited), T_cw, K, on_device=True)
    steps = win.build(proj_dev, T_cw, cam_azi, 640, 480).cpu().numpy()
    assert np.array_equal(one, steps)
    win.close()


def test_stores_created_while_the_default_stream_is_busy():
    """A caller may run the association on a stream of its own (torch.cuda.stream) while other work sits on the default stream --
    bench.py's overlapped mode does, with the detector in another thread.  Buffers the library zeroes when it CREATES a store must be
    zero before the create call returns: hipMemset is ordered on the NULL stream, and a memset still queued there behind foreign
    work landed after the caller's first load on its own stream (track counts back to zero -> windows of -1).  Track store and
    association handle created and used on a side stream behind ~100 ms of work on the default stream, against the default-stream run."""
    import torch
    from odam_amd import associator, weights
    from odam_amd.associator import TrackWindows
    from odam_amd.processor import OdamProcess, get_cam_azi
    rs = np.random.RandomState(5)
    proc = OdamProcess(None, None, None, None)
    proc.init_sequence(np.array([[577.87, 0, 319.5], [0, 577.87, 239.5], [0, 0, 1.0]]), 480, 640)
    Q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
    T_wc = np.eye(4); T_wc[:3, :3] = Q; T_wc[:3, 3] = rs.normal(size=3)
    cam_azi = get_cam_azi(T_wc)
    tracks = _random_tracks(rs, [3, 120, 40, 7])
    proj = rs.uniform(-200, 900, (len(tracks), 4))
    want = np.ascontiguousarray(proc._preprocess_tracks(tracks, T_wc, cam_azi, proj_px=proj).transpose(0, 2, 1))
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "assoc.npz"))
    tr, de, Zr = z["c0_tracks"], z["c0_det"], z["c0_Z"]
    T, n_det = Zr.shape[0] - 1, Zr.shape[1] - 1
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    sd = weights.make_associator_state_dict(2, 8, seed=0)
    a0 = associator.Associator(cfg, max_tracks=64, device=DEV); a0.load_state_dict(sd)
    Z0 = a0.assignment_checked(torch.from_numpy(tr), torch.from_numpy(de), T, n_det).cpu().numpy()
    a0.close()
    busy = torch.randn(8192, 8192, device=DEV)
    side = torch.cuda.Stream()
    for _ in range(30):
        busy = busy @ busy * 1e-4                     # queued on the default stream
    with torch.cuda.stream(side):
        win = TrackWindows(DEV)
        win.load(tracks)
        got = win.build(torch.from_numpy(proj).to(DEV), np.linalg.inv(T_wc), cam_azi, 640, 480)
        a1 = associator.Associator(cfg, max_tracks=64, device=DEV); a1.load_state_dict(sd)
        Z1 = a1.assignment_checked(torch.from_numpy(tr), torch.from_numpy(de), T, n_det)
        side.synchronize()
        got, Z1 = got.cpu().numpy(), Z1.cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(got, want)
    assert np.array_equal(Z1, Z0)
    win.close(); a1.close()


def test_row_partitioned_matching_kernel_equals_the_device_wide_one(golden):
    """odam_config assoc.persist = 2 (default): the matching GNN with the rows dealt to the XCDs -- one device-wide barrier per layer, XCD-local
    barriers and plain stores for the row-wise stages (csrc/assoc.hip gnn_rowpart_kernel) -- against assoc.persist = 1, five device-wide
    barriers per layer: every output element is computed by the same instructions in the same order, so the assignment matrices are equal
    bit for bit.  Reference-run cases plus row counts around the partition's edges (1 ... 157 rows: XCDs without rows, several 16-row
    blocks per XCD)."""
    import torch
    from odam_amd import _lib, associator, weights
    z = golden("assoc.npz")
    cfg = {"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"], "sinkhorn_iterations": 100}
    a = associator.Associator(cfg, max_tracks=128, device=DEV); a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
    rng = np.random.default_rng(0)
    cases = [(z[f"c{c}_tracks"], z[f"c{c}_det"], z[f"c{c}_Z"].shape[0] - 1, z[f"c{c}_Z"].shape[1] - 1) for c in range(int(z["n_cases"]))]
    for T in (1, 7, 8, 9, 63, 64, 65, 100, 127):
        cases.append((rng.standard_normal((T, 79, 100)).astype(np.float32), rng.standard_normal((79, 30)).astype(np.float32), T, int(rng.integers(1, 31))))
    try:
        for tr, de, T, n in cases:
            Z = {}
            for v in (1, 2):
                _lib.set_config("assoc.persist", v)
                Z[v] = a.assignment_checked(torch.from_numpy(tr), torch.from_numpy(de), T, n).cpu().numpy()
            assert np.isfinite(Z[2]).all()
            assert np.array_equal(Z[1].view(np.uint32), Z[2].view(np.uint32)), (T, n)
        assert a.lost_launches == 0
    finally:
        _lib.set_config("assoc.persist", 2)
        a.close()


def _hungarian_device(scores_or_Z, thr, log_domain):
    """odam_assoc_hungarian on a [T, n] float32 matrix -> match [n] (float64, as hungarian_matching returns), status"""
    from odam_amd import _lib
    M = torch.as_tensor(scores_or_Z, dtype=torch.float32).contiguous().to(DEV)
    T, n = M.shape
    out = torch.full((32,), -7, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().odam_assoc_hungarian(_lib.ptr(M), ctypes.c_int(T), ctypes.c_int(n), ctypes.c_int(n), ctypes.c_double(thr),
                                               ctypes.c_int(log_domain), _lib.ptr(out), ctypes.c_void_p(out.data_ptr() + 4 * 31),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "odam_assoc_hungarian")
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[:n].astype(np.float64), int(o[31])


def test_hungarian_on_the_device_equals_scipy():
    """odam_assoc_hungarian (scipy's rectangular shortest-augmenting-path solver restated on one wavefront, associator.py:19-35) against
    hungarian_matching = scipy.optimize.linear_sum_assignment on the same float32 scores: random matrices wide, tall and square up to
    30 detections x 128 tracks, matrices FULL of exact ties (constants, few distinct values, duplicated rows and columns: scipy's tie order
    is part of the result), and every log-assignment matrix of the two reference-run fixtures (there through the device's own exp)."""
    from odam_amd.associator import hungarian_matching
    rng = np.random.default_rng(5)
    cases = []
    for T, n in ((5, 3), (3, 5), (30, 30), (42, 17), (17, 30), (100, 30), (128, 30), (1, 1), (1, 30), (60, 1), (31, 2), (2, 31)):
        for rep in range(3):
            cases.append(rng.random((T, n), dtype=np.float32))
    for T, n in ((6, 6), (9, 4), (4, 9), (40, 30), (30, 12)):
        cases.append(np.full((T, n), 0.5, np.float32))                                   # scipy: the identity on a constant matrix
        cases.append((rng.integers(0, 3, (T, n)) / 4).astype(np.float32))                # three distinct values
        cases.append((rng.integers(0, 8, (T, n)) / 8).astype(np.float32))
        a = rng.random((T, n), dtype=np.float32); a[1::2] = a[0::2][:len(a[1::2])]       # duplicated rows
        cases.append(a)
        b = rng.random((T, n), dtype=np.float32); b[:, 1::2] = b[:, 0::2][:, :b[:, 1::2].shape[1]]      # duplicated columns
        cases.append(b)
    for thr in (0.0, 0.2, 0.6):
        for sc in cases:
            want = hungarian_matching(torch.from_numpy(sc), thr)
            got, st = _hungarian_device(sc, thr, 0)
            assert st == 0 and np.array_equal(got, want), (sc.shape, thr, got, want)
    n_fix = 0
    for name in ("e2e.npz", "process_tracks_net.npz"):
        z = np.load(os.path.join(GOLD, name), allow_pickle=True)
        for k in [k for k in z.files if k.startswith("Z") and k[1:].isdigit()]:
            Z = torch.from_numpy(z[k].astype(np.float32))
            if Z.shape[0] < 2 or Z.shape[1] < 2:
                continue
            for thr in (0.1, 0.2):
                want = hungarian_matching(Z[:-1, :-1].exp(), thr)
                got, st = _hungarian_device(Z[:-1, :-1].contiguous(), thr, 1)
                assert st == 0 and np.array_equal(got, want), (name, k, thr)
                n_fix += 1
    assert n_fix >= 100
    # beyond 32 x 128 the entry refuses (the host then solves it), and an all-NaN matrix (an abandoned persistent launch) is reported as infeasible
    from odam_amd import _lib
    big = torch.rand(40, 33, device=DEV)
    out = torch.zeros(64, dtype=torch.int32, device=DEV)
    assert _lib.lib().odam_assoc_hungarian(_lib.ptr(big), 40, 33, 33, ctypes.c_double(0.1), 0, _lib.ptr(out), ctypes.c_void_p(out.data_ptr() + 4 * 63),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 3
    _, st = _hungarian_device(np.full((4, 3), np.nan, np.float32), 0.1, 0)
    assert st == 1

