"""Golden for the WHOLE chain the reference's driver executes (likojack/ODAM src/scripts/run_processor.py:70-83):

    frames -> OdamProcess.process_frame (DETR detector -> post-process + NMS -> detection rows -> track preparation ->
    Associator -> attach)  x 40   ->   optim_process  ->  merge_process  ->  optim_process

run with the reference's own modules on the CPU: src/processor.py, src/models/{detr,transformer,position_encoding,
backbone,associator}.py, src/scripts/{run_multi_view,run_merge}.py, src/super_quadric/*.  Two things are not the
reference's: the ResNet body (torchvision is neither vendored nor installed: the stand-in module of
make_golden_detr.py, "parity unpinned" for that part) and the weights (synthetic, odam_amd/weights.py scene=True for
the detector, make_associator_state_dict for the associator -- the trained checkpoints are not available offline).
Frames are 256x320 and the caller-supplied transform has size=256 (no resize), which keeps the CPU run short.

Stored (tests/golden/e2e.npz): per-frame detection rows, the tracks after the 40 frames, fitted parameters / boxes of
the first pass, the merged tracks, fitted parameters / boxes of the second pass; and, for both fit passes, the
reference's own outputs (a) when one world-position column of every track is scaled by (1 +- k 2^-23) (six runs each:
its ulp-level self-sensitivity, the yardstick for a fit given IDENTICAL inputs) and (b) when every continuous track
column (box, dimensions, position, azimuth) carries relative noise of 2e-5 (six seeds: the size by which two correct
fp32 detectors differ, measured 1.2e-5 median / 1.9e-5 max between the GPU and this CPU run -- the yardstick for the
fit at the end of the whole chain).  Also stored: the assignment matrix of every association call (Z<i>), which the GPU
test compares frame by frame together with the decisions taken from it.  Inputs are regenerated from seeds
(odam_amd.synth.make_sequence).
Run: python tests/golden/make_golden_e2e.py            (about 3 minutes on 8 cores)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SEQ = dict(n=40, h=256, w=320, seed=11)
NUDGES = [(9, 1), (10, -1), (11, 1), (9, -2), (10, 2), (11, -2)]
NOISE = 2e-5


def _spread(job):
    """one nudged optim_process of the reference over stored tracks"""
    which, k = job
    torch.set_num_threads(1)
    import refenv
    refenv.setup()
    from make_golden import params_of
    from src.scripts.run_multi_view import optim_process
    from odam_amd import synth
    z = np.load(os.path.join(HERE, "e2e.npz"))
    seq = synth.make_sequence(**SEQ)
    name, n = ("track", int(z["n_tracks"])) if which == 1 else ("merged", int(z["n_merged"]))
    tracks = []
    rs = np.random.RandomState(100 + k)
    for i in range(n):
        t = z[f"{name}{i}"].copy()
        if k < len(NUDGES):
            col, m = NUDGES[k]
            t[:, col] *= 1.0 + m * 2.0 ** -23
        else:
            t[:, 2:13] *= 1.0 + NOISE * rs.uniform(-1, 1, t[:, 2:13].shape)
            t[:, 78:82] = t[:, 2:6]
        tracks.append(t)
    P_cws = [seq["K"] @ np.linalg.inv(T)[:3, :] for T in seq["T_wcs"]]
    out = optim_process(tracks, seq["img_names"], list(seq["T_wcs"]), P_cws, SEQ["h"], SEQ["w"], seq["K"], "super_quadric",
                        True, 200, 10)
    return which, k, np.stack([params_of(q) for q in out["quadrics"]]), np.asarray(out["bboxes_qc"])

SCORE_THRESHOLD = 0.8      # the reference default (processor.py:24)


def main():
    import refenv
    refenv.setup()
    import torchvision
    from make_golden_detr import _ResNet
    from make_golden import params_of, patch_run_merge
    from make_golden_assoc import CFG
    torchvision.models.resnet50 = lambda replace_stride_with_dilation=None, pretrained=False, norm_layer=None: _ResNet((3, 4, 6, 3), norm_layer)
    torch.Tensor.cuda = lambda self, *a, **k: self          # DETR.postprocess hard-codes .cuda() (detr.py:113-133)
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)
    from PIL import Image
    from src.config.configs import ConfigLoader
    from src.models.detr import build as build_detector
    from src.models.associator import Associator
    import src.processor as ref_processor
    from odam_amd import weights, synth, transforms
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import scene_weights
    rm = patch_run_merge()
    ref_processor.merge_process = rm.merge_process

    cfg = ConfigLoader().merge_cfg(["/root/reference/configs/detr_scan_net.yaml"])
    cfg.device = "cpu"
    model, _, _ = build_detector(cfg)
    missing = model.load_state_dict(weights.make_state_dict(seed=0, scene=True), strict=False)
    assert not missing.unexpected_keys and all(k.startswith("backbone.0.body.fc") for k in missing.missing_keys), missing
    model.eval()
    net = Associator(CFG)
    net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0), strict=True)
    net.eval()
    fwd = net.forward
    zs = []

    def forward(data, thr, eval_only=False):
        out = fwd(data, thr, eval_only=eval_only, device="cpu")
        zs.append(out["pred"][0][0].detach().numpy().astype(np.float32))
        return out
    net.forward = forward

    seq = synth.make_sequence(**SEQ)
    proc = ref_processor.OdamProcess(model, net, transforms.Transforms(size=SEQ["h"]), None, score_threshold=SCORE_THRESHOLD)
    proc.init_sequence(seq["K"], SEQ["h"], SEQ["w"])
    dets = {}
    run_detector = proc.run_detector

    def logged(rgb, fid, T):
        with torch.no_grad():
            rows = run_detector(rgb, fid, T)
        dets[fid] = np.asarray(rows, np.float64).reshape(-1, 79)
        return rows
    proc.run_detector = logged
    for i, fid in enumerate(seq["img_names"]):
        proc.process_frame(Image.fromarray(seq["frames"][i]), fid, seq["T_wcs"][i])
        print("frame", i, "detections", len(dets[fid]), "tracks", len(proc.tracks), flush=True)
    tracks = [t.copy() for t in proc.tracks]
    out1 = proc.optim_process([t.copy() for t in tracks])
    merged = proc.merge_process({"tracks": [t.copy() for t in out1["tracks"]], "bboxes_qc": out1["bboxes_qc"]})
    merged_in = [t.copy() for t in merged]
    out2 = proc.optim_process([t.copy() for t in merged])
    torch.Tensor.to = orig_to

    d = dict(n_tracks=np.int32(len(tracks)), n_merged=np.int32(len(merged_in)), score_threshold=np.float64(SCORE_THRESHOLD),
             params1=np.stack([params_of(q) for q in out1["quadrics"]]), bboxes_qc1=np.asarray(out1["bboxes_qc"]),
             bboxes_dl1=np.asarray(out1["bboxes_dl"]),
             params2=np.stack([params_of(q) for q in out2["quadrics"]]), bboxes_qc2=np.asarray(out2["bboxes_qc"]),
             bboxes_dl2=np.asarray(out2["bboxes_dl"]))
    for i, t in enumerate(tracks):
        d[f"track{i}"] = t
    for i, t in enumerate(merged_in):
        d[f"merged{i}"] = t
    for fid, r in dets.items():
        d[f"det{fid}"] = r
    d["n_calls"] = np.int32(len(zs))
    for i, zz in enumerate(zs):
        d[f"Z{i}"] = zz
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **d)
    print("e2e golden:", len(tracks), "tracks (lengths", [len(t) for t in tracks], ") ->", len(merged_in), "merged (lengths",
          [len(t) for t in merged_in], "); classes", [int(np.median(t[:, 1])) for t in tracks])
    spreads()


def spreads():
    from multiprocessing import Pool
    d = dict(np.load(os.path.join(HERE, "e2e.npz")))
    nn = len(NUDGES)
    with Pool(6) as pool:
        res = pool.map(_spread, [(w, k) for w in (1, 2) for k in range(2 * nn)], chunksize=1)
    for w in (1, 2):
        rs = sorted([r for r in res if r[0] == w], key=lambda r: r[1])
        d[f"spread_params{w}"] = np.stack([r[2] for r in rs[:nn]]); d[f"spread_bboxes_qc{w}"] = np.stack([r[3] for r in rs[:nn]])
        d[f"noise_params{w}"] = np.stack([r[2] for r in rs[nn:]]); d[f"noise_bboxes_qc{w}"] = np.stack([r[3] for r in rs[nn:]])
    d["nudges"] = np.asarray(NUDGES, np.int32); d["noise"] = np.float64(NOISE)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **d)



if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "spreads":      # only the sensitivity runs, over the stored tracks
        spreads()
    else:
        main()
