"""gfx950 association network (GNN + fused Sinkhorn) against the reference Associator's outputs
(tests/golden/assoc.npz, made by importing src/models/associator.py on the CPU)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


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


def test_forward_vs_reference_associator(golden):
    from odam_amd import associator, weights
    z = golden("assoc.npz")
    a = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                               "sinkhorn_iterations": 100}, max_tracks=64, device=DEV)
    a.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))
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


def test_process_sequence_with_network(golden):
    """The whole per-frame loop (canned detections -> _prepare_tracks -> association network -> Hungarian ->
    track attachment) on the GPU against the reference's OdamProcess + reference Associator run on the CPU:
    the same tracks, i.e. identical object IDs for every detection of the 40-frame sequence."""
    from odam_amd import associator, sq, weights
    from odam_amd.processor import OdamProcess
    z = golden("process_tracks.npz")
    ref = golden("process_tracks_net.npz")
    net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                                 "sinkhorn_iterations": 100}, max_tracks=64, device=DEV)
    net.load_state_dict(weights.make_associator_state_dict(2, 8, seed=0))

    class _Det:
        device = DEV
    proc = OdamProcess(_Det(), net, None, None, score_threshold=0.2, fitter=sq.SqFitter(DEV, 1))
    proc.init_sequence(z["K"], 480, 640)
    for i, fid in enumerate(z["img_names"]):
        proc.process_frame(None, int(fid), z["scene_T_wcs"][i], detections=[list(r) for r in z[f"det{int(fid)}"]])
    assert len(proc.tracks) == int(ref["n_tracks"])
    for i, t in enumerate(proc.tracks):
        r = ref[f"track{i}"]
        assert t.shape == r.shape, i
        assert np.array_equal(t[:, :2], r[:, :2])
        assert np.allclose(t, r, rtol=1e-6, atol=1e-6)
    net.close()
