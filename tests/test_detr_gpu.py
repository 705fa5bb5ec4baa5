"""gfx950 detector kernels and the full DETR forward (through the C ABI) against torch CPU references,
the detector oracle and the reference-generated golden vectors."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "oracle"))

pytestmark = pytest.mark.gpu
K = np.array([[577.87, 0.0, 319.5], [0.0, 577.87, 239.5], [0.0, 0.0, 1.0]])
KEYS = ("pred_logits", "pred_boxes", "pred_angle", "pred_offset", "pred_size", "pred_depth")
DEV = "cuda:0"


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def _pack(w, epc=4, k_order=0):
    """[Cout,Cin,KH,KW] -> [Cout][Kpad] in the kernel's K order (include/odam_detr.h): 0 = (tap, ci),
    1 = (ci // kt, tap, ci % kt) with kt = 8 chunks of `epc` elements"""
    Cout, Cin, KH, KW = w.shape
    kt = 8 * epc
    CinP = (Cin + epc - 1) // epc * epc
    wp = torch.zeros(Cout, KH, KW, CinP); wp[..., :Cin] = w.permute(0, 2, 3, 1)
    Kk = KH * KW * CinP; Kpad = (Kk + kt - 1) // kt * kt
    if k_order:
        assert CinP % kt == 0
        wp = wp.reshape(Cout, KH * KW, CinP // kt, kt).permute(0, 2, 1, 3)
    out = torch.zeros(Cout, Kpad); out[:, :Kk] = wp.reshape(Cout, Kk)
    return out, CinP, Kpad


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,relu,res", [
    (2, 37, 45, 64, 64, 1, 1, 0, True, False),      # layer1 1x1, N = 64 tile
    (1, 40, 53, 64, 64, 3, 1, 1, True, False),      # 3x3 stride 1, ragged M
    (2, 33, 41, 128, 128, 3, 2, 1, True, False),    # 3x3 stride 2, odd sizes (layer2.0)
    (1, 80, 106, 3, 64, 7, 2, 3, True, False),      # stem 7x7/2, Cin 3 -> 4, K 196 -> 224
    (2, 25, 34, 256, 1024, 1, 1, 0, True, True),    # expand + residual + relu
    (2, 50, 67, 512, 1024, 1, 2, 0, False, False),  # strided 1x1 downsample
    (1, 1, 300, 256, 19, 1, 1, 0, False, False),    # class head: N = 19
    (3, 25, 34, 2048, 256, 1, 1, 0, False, False),  # input_proj, K = 2048
    (1, 1, 1, 256, 1, 1, 1, 0, False, False),       # depth head on one token
    (5, 5, 5, 32, 48, 3, 1, 1, False, False),       # a 128-row tile spans six images; one k-tile per filter tap
    (7, 6, 9, 64, 96, 5, 1, 2, True, False),        # 25 taps: the widest kernel the tap bitmask path takes
    (2, 12, 12, 32, 32, 7, 1, 3, False, False),     # 49 taps: register-staged gather with Cin >= 32
    (3, 9, 9, 16, 32, 3, 1, 1, True, False),        # Cin < k-tile: two taps inside one k-tile
    (2, 31, 29, 64, 256, 3, 2, 1, False, True),     # stride 2 + residual, N = 256
    (3, 25, 34, 256, 400, 1, 1, 0, True, True),     # N = 400: the second 256-column tile ends inside its second 128-column half (ring kernel, cg.ring=2)
    (2, 23, 31, 128, 644, 1, 1, 0, False, True),    # N = 644: three column tiles, the last one 132 columns wide
])
def test_conv_gemm_vs_torch(B, H, W, Cin, Cout, k, s, p, relu, res):
    from odam_amd import _lib
    L = _lib.lib()
    torch.manual_seed(B * 1000 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    sc = torch.rand(Cout) + 0.5; bi = torch.randn(Cout)
    y = F.conv2d(x, w, None, s, p) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)
    r = torch.randn_like(y) if res else None
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    for k_order in ((0, 1) if Cin % 32 == 0 and k * k <= 32 else (0,)):
        wp, CinP, Kpad = _pack(w, 4, k_order)
        xh = torch.zeros(B, H, W, CinP); xh[..., :Cin] = x.permute(0, 2, 3, 1)
        dx, dw, dsc, dbi = xh.to(DEV), wp.to(DEV), sc.to(DEV), bi.to(DEV)
        dr = r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
        dy = torch.empty(B, y.shape[2], y.shape[3], Cout, device=DEV)
        _lib.check(L.odam_op_conv2d_nhwc(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dsc), _lib.ptr(dbi), _lib.ptr(dr), _lib.ptr(dy),
                                         B, H, W, CinP, Cout, k, k, s, p, Kpad, int(relu), k_order, _st()), "conv")
        got = dy.cpu().permute(0, 3, 1, 2)
        assert (got - y).abs().max().item() <= 3e-6 * max(y.abs().max().item(), 1.0), k_order   # fp32, different sum order


@pytest.mark.parametrize("M,Cin,Cout,relu,res", [
    (200, 256, 256, False, True),      # a decoder projection of two frames
    (700, 256, 2048, True, False),     # the decoder's first feed-forward layer, seven frames
    (3200, 2048, 256, False, True),    # ... the second one of a full batch: the tile choice flips to 128 x 128 between these row counts
    (131, 128, 40, True, False),       # ragged rows and columns
])
def test_small_tile_shapes_are_bit_identical(M, Cin, Cout, relu, res):
    """What pipeline.py's N-rank == 1-rank bit identity rests on beside cg.pin (conv_gemm.hip, the small-tile fall-through still picks
    64 x 64 or 128 x 128 tiles by the number of tiles, i.e. by M): every small-tile shape accumulates an output element's K in the same
    order (k-tiles ascending, one matrix instruction chain per element), so the SHAPE never shows in the result.  Stated here as a
    test: cg.force 1 / 2 / 3 (128 x 64, 64 x 64, 128 x 128) and the library's own choice return the same bits, fp32 and bf16."""
    from odam_amd import _lib
    L = _lib.lib()
    torch.manual_seed(M + Cout)
    x = torch.randn(1, Cin, 1, M); w = torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5
    sc = (torch.rand(Cout) + 0.5).to(DEV); bi = torch.randn(Cout).to(DEV)
    r = torch.randn(1, 1, M, Cout) if res else None
    try:
        for bf in (False, True):
            wp, CinP, Kpad = _pack(w, 8 if bf else 4, 0)
            xh = torch.zeros(1, 1, M, CinP); xh[..., :Cin] = x.permute(0, 2, 3, 1)
            d = (lambda t: _to_bf16_bits(t.contiguous()).to(DEV)) if bf else (lambda t: t.contiguous().to(DEV))
            dx, dw, dr = d(xh), d(wp), (d(r) if res else None)
            outs = []
            for force in (0, 1, 2, 3):
                if force == 1 and Cout > 64:
                    continue
                _lib.set_config("cg.force", force)
                dy = torch.full((1, 1, M, Cout), -1, device=DEV, dtype=torch.int16) if bf else torch.full((1, 1, M, Cout), float("nan"), device=DEV)
                if bf:
                    _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(sc), _lib.ptr(bi), _lib.ptr(dr), _lib.ptr(dy),
                                                          1, 1, M, CinP, Cout, 1, 1, 1, 0, Kpad, int(relu), 0, 0, _st()), "conv bf16")
                else:
                    _lib.check(L.odam_op_conv2d_nhwc(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(sc), _lib.ptr(bi), _lib.ptr(dr), _lib.ptr(dy),
                                                     1, 1, M, CinP, Cout, 1, 1, 1, 0, Kpad, int(relu), 0, _st()), "conv")
                torch.cuda.synchronize()
                outs.append((force, dy.clone()))
            for force, o in outs[1:]:
                assert torch.equal(o, outs[0][1]), (bf, force, (o != outs[0][1]).float().mean().item())
    finally:
        _lib.set_config("cg.force", 0)


@pytest.mark.parametrize("B,Lq,Lk", [(2, 850, 850), (3, 100, 100), (2, 100, 850), (1, 33, 70), (1, 1, 1)])
def test_attention_vs_torch(B, Lq, Lk):
    from odam_amd import _lib
    torch.manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, 256); k = torch.randn(B, Lk, 256); v = torch.randn(B, Lk, 256)
    qh = q.view(B, Lq, 8, 32).transpose(1, 2) * (1.0 / 32) ** 0.5
    kh = k.view(B, Lk, 8, 32).transpose(1, 2); vh = v.view(B, Lk, 8, 32).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(B, Lq, 256)
    dq, dk, dv = q.to(DEV), k.to(DEV), v.to(DEV)
    do = torch.empty(B, Lq, 256, device=DEV)
    _lib.check(_lib.lib().odam_op_attention(_lib.ptr(dq), 256, _lib.ptr(dk), 256, _lib.ptr(dv), 256, _lib.ptr(do), 256,
                                            B, 8, Lq, Lk, _st()), "attention")
    assert (do.cpu() - ref).abs().max().item() <= 5e-6


def test_layernorm_and_maxpool_vs_torch():
    from odam_amd import _lib
    L = _lib.lib()
    torch.manual_seed(0)
    x = torch.randn(777, 256); r = torch.randn(777, 256); g = torch.rand(256) + 0.5; b = torch.randn(256)
    ref = F.layer_norm(x + r, (256,), g, b, 1e-5)
    dx, dr, dg, db = x.to(DEV), r.to(DEV), g.to(DEV), b.to(DEV)
    dy = torch.empty(777, 256, device=DEV)
    _lib.check(L.odam_op_add_layernorm(_lib.ptr(dx), _lib.ptr(dr), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dy), 777, _st()), "ln")
    assert (dy.cpu() - ref).abs().max().item() <= 5e-6
    x = torch.randn(2, 64, 41, 53)
    ref = F.max_pool2d(x, 3, 2, 1)
    dx = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dy = torch.empty(2, ref.shape[2], ref.shape[3], 64, device=DEV)
    _lib.check(L.odam_op_maxpool3x3s2_nhwc(_lib.ptr(dx), _lib.ptr(dy), 2, 41, 53, 64, _st()), "pool")
    assert torch.equal(dy.cpu().permute(0, 3, 1, 2), ref)


@pytest.fixture(scope="module")
def model():
    from odam_amd import detector, weights
    sd = weights.make_state_dict(seed=0)
    det = detector.Detector(max_batch=4, device=DEV)
    det.load_state_dict(sd)
    yield det, sd
    det.close()


def test_forward_vs_reference_golden(model, golden):
    """the reference's own DETR output (imported in the build container) at 2 x 3x256x320"""
    det, _ = model
    z = golden("detr_small.npz")
    torch.manual_seed(int(z["img_seed"]))
    img = torch.randn(2, 3, 256, 320)
    out = det(img.to(DEV))
    for k in KEYS:
        assert np.abs(out[k].cpu().numpy() - z[k]).max() <= 1e-4, k
    assert np.array_equal(out["pred_logits"].argmax(-1).cpu().numpy(), z["pred_logits"].argmax(-1))   # class labels: exact
    post = det.postprocess(out, (640, 480), 0.6, K)
    for b in range(2):
        assert np.array_equal(post["classes"][b], z[f"post{b}_classes"])
        for k in ("bboxes", "dimensions", "angles", "translates", "scores"):
            assert np.allclose(post[k][b], z[f"post{b}_{k}"], rtol=1e-4, atol=1e-3), (b, k)


@pytest.mark.parametrize("variant", ["pre", "learned", "pre_learned", "dc5"])
def test_config_variants_vs_reference_golden(golden, variant):
    """pre_norm / learned position embedding / dilated layer4 (the config variants besides the shipped one):
    the GPU forward against the reference's own DETR built with those keys (detr_variants.npz), and build(cfg) accepts them"""
    from odam_amd import detector, weights
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_detr_variants import image
    z = golden("detr_variants.npz")
    det, _, _ = detector.build({"pre_norm": variant.startswith("pre"), "position_embedding": "learned" if variant.endswith("learned") else "sine",
                                "dilation": variant == "dc5"})
    det.load_state_dict(weights.add_variant_weights(weights.make_state_dict(seed=0)))
    out = det(image().to(DEV))
    for k in KEYS:
        assert np.abs(out[k].cpu().numpy() - z[f"{variant}_{k}"]).max() <= 1e-4 * max(1.0, float(np.abs(z[f"{variant}_{k}"]).max())), k
    assert np.array_equal(out["pred_logits"].argmax(-1).cpu().numpy(), z[f"{variant}_pred_logits"].argmax(-1))
    det.close()


def test_full_size_forward_vs_reference_golden(model, golden):
    """the reference's own DETR output on one 3x800x1066 frame (850 encoder tokens): BASELINE's full size"""
    det, _ = model
    z = golden("detr_full.npz")
    torch.manual_seed(int(z["img_seed"]))
    img = torch.randn(1, 3, 800, 1066)
    out = det(img.to(DEV))
    for k in KEYS:
        assert np.abs(out[k].cpu().numpy() - z[k]).max() <= 1e-4 * max(1.0, float(np.abs(z[k]).max())), k
    assert np.array_equal(out["pred_logits"].argmax(-1).cpu().numpy(), z["pred_logits"].argmax(-1))   # class labels: exact


def test_full_size_forward_vs_oracle(model):
    """BASELINE config size (800x1066 from a 640x480 frame): taps + outputs + discrete decisions vs the oracle"""
    import detr_oracle as O
    det, sd = model
    torch.manual_seed(7)
    img = torch.randn(2, 3, 800, 1066)
    ref = O.detr_forward(sd, img, return_taps=True)
    out = det(img.to(DEV))
    l4, mem = det.debug_taps(2, 800, 1066)
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    assert rel(l4.cpu(), ref["_layer4"]) <= 2e-5        # 53 stacked fp32 convolutions
    assert rel(mem.cpu(), ref["_memory"]) <= 2e-5
    for k in KEYS + ("pred_obj_features",):
        assert (out[k].cpu() - ref[k]).abs().max().item() <= 2e-4, k
    pc = out["pred_logits"].cpu().softmax(-1)[..., :-1]; pr = ref["pred_logits"].softmax(-1)[..., :-1]
    assert torch.equal(pc.argmax(-1), pr.argmax(-1))
    assert torch.equal(pc.max(-1).values > 0.6, pr.max(-1).values > 0.6)
    pp = det.postprocess(out, (640, 480), 0.6, K); pref = O.postprocess(ref, (640, 480), 0.6, K)
    for b in range(2):
        assert np.array_equal(pp["classes"][b], pref["classes"][b])
        assert np.allclose(pp["translates"][b], pref["translates"][b], rtol=1e-4, atol=1e-4)


def test_full_size_all_variants_vs_oracle():
    """pre_norm + learned positions at the full size (25 x 34 tokens), and the dilated layer4 at a size where its 3x3 launches
    run on the 256-row ring kernel (dilation in the tap offsets of the LDS-DMA gather) with 40 x 50 = 2,000 tokens"""
    import detr_oracle as O
    from odam_amd import detector, weights
    sd = weights.add_variant_weights(weights.make_state_dict(seed=0))
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    from odam_amd import _lib
    ring = _lib.config()["cg.ring"]
    for kw, okw, (B, H, W) in ((dict(pre_norm=True, position_embedding="learned"), dict(pre_norm=True, learned_pos=True), (1, 800, 1066)),
                               (dict(dilation=True), dict(dilation=True), (2, 640, 800))):
        _lib.set_config("cg.ring", 2 if "dilation" in kw else ring)      # the ring kernel on every eligible layer: two frames are too few tiles for the default rule
        det = detector.Detector(max_batch=B, device=DEV, n_streams=1, **kw); det.load_state_dict(sd)
        torch.manual_seed(11)
        img = torch.randn(B, 3, H, W)
        ref = O.detr_forward(sd, img, return_taps=True, **okw)
        out = det(img.to(DEV))
        l4, mem = det.debug_taps(B, H, W)
        assert rel(l4.cpu(), ref["_layer4"]) <= 2e-5, kw
        assert rel(mem.cpu(), ref["_memory"]) <= 3e-5, kw
        for k in KEYS:
            assert (out[k].cpu() - ref[k]).abs().max().item() <= 2e-4 * max(1.0, ref[k].abs().max().item()), (kw, k)
        assert torch.equal(out["pred_logits"].cpu().argmax(-1), ref["pred_logits"].argmax(-1)), kw
        det.close()
    _lib.set_config("cg.ring", ring)


def test_batch_size_does_not_change_results(model):
    det, _ = model
    torch.manual_seed(9)
    img = torch.randn(4, 3, 256, 320, device=DEV)
    a = det(img)
    b = [det(img[i:i + 1]) for i in range(4)]
    # to fp32 reorder noise, not bit for bit: kernels and tile shapes are chosen by problem size (four frames of this size
    # take conv1 through the ring kernel, one frame does not); frames inside batches of one size agree exactly
    for k in KEYS:
        d = (a[k] - torch.cat([x[k] for x in b])).abs().max().item()
        assert d <= 2e-5 * max(1.0, a[k].abs().max().item()), (k, d)
    c = det(img)
    for k in KEYS:
        assert torch.equal(a[k], c[k]), k


def test_postprocess_rows_vs_reference_golden(model, golden):
    det, _ = model
    z = golden("detr_post.npz")
    out = {k: torch.from_numpy(z[k]).to(DEV) for k in KEYS}
    out["_hw"] = (256, 320)
    det(torch.zeros(1, 3, 256, 320, device=DEV))  # make sure the handle for this size exists
    post = det.postprocess(out, (640, 480), 0.6, K)
    for b in range(4):
        assert np.array_equal(post["classes"][b], z[f"post{b}_classes"])
        for k in ("bboxes", "dimensions", "angles", "translates", "scores"):
            assert np.allclose(post[k][b], z[f"post{b}_{k}"], rtol=1e-5, atol=2e-4), (b, k)


def test_run_detector_rows_vs_reference_golden(model, golden):
    from odam_amd import synth, transforms
    from odam_amd.processor import OdamProcess
    det, _ = model
    ref = golden("detr_small.npz")["run_detector_rows"]
    proc = OdamProcess(det, None, transforms.get_transforms(), None)
    proc.init_sequence(K, 968, 1296)
    from PIL import Image
    rows = np.asarray(proc.run_detector(Image.fromarray(next(synth.make_frames(1, seed=5))), 7, np.eye(4))).reshape(-1, 79)
    assert rows.shape == ref.shape
    assert np.array_equal(rows[:, :2], ref[:, :2])
    assert np.allclose(rows, ref, rtol=1e-4, atol=1e-4)


def test_resnet101_configuration_vs_oracle():
    """BASELINE config 4 backbone (R101) at a reduced input"""
    import detr_oracle as O
    from odam_amd import detector, weights
    sd = weights.make_state_dict(backbone="resnet101", seed=1)
    det = detector.Detector(backbone="resnet101", max_batch=2, device=DEV)
    det.load_state_dict(sd)
    torch.manual_seed(3)
    img = torch.randn(2, 3, 192, 256)
    ref = O.detr_forward(sd, img, blocks=(3, 4, 23, 3))
    out = det(img.to(DEV))
    for k in KEYS:
        assert (out[k].cpu() - ref[k]).abs().max().item() <= 3e-4, k
    assert torch.equal(out["pred_logits"].cpu().argmax(-1), ref["pred_logits"].argmax(-1))
    det.close()


@pytest.mark.parametrize("B,H,W,bb", [(3, 601, 795, "resnet50"), (2, 487, 1013, "resnet101")])
def test_large_odd_sizes_vs_oracle(B, H, W, bb):
    """Odd extents at sizes where the ring kernel, conv1-as-rows, the split attention and the fused layer1 / layer2
    bottlenecks (incl. the chain into the next reduce) all apply: ragged last tiles in every one of them."""
    import detr_oracle as O
    from odam_amd import detector, weights
    sd = weights.make_state_dict(backbone=bb, seed=0, scene=True)
    det = detector.Detector(backbone=bb, max_batch=B, device=DEV, n_streams=1)
    det.load_state_dict(sd)
    torch.manual_seed(H + W)
    img = torch.randn(B, 3, H, W)
    torch.set_num_threads(8)
    ref = O.detr_forward(sd, img, blocks=(3, 4, 23, 3) if bb == "resnet101" else (3, 4, 6, 3))
    got = det(img.to(DEV))
    for k in KEYS:
        assert (got[k].cpu() - ref[k]).abs().max().item() <= 2e-4 * max(1.0, ref[k].abs().max().item()), k
    assert torch.equal(got["pred_logits"].cpu().argmax(-1), ref["pred_logits"].argmax(-1))
    det.close()


@pytest.mark.parametrize("B,H,W", [(2, 65, 97), (3, 32, 33), (1, 131, 77)])
def test_odd_input_sizes_vs_oracle(B, H, W):
    """sizes that leave ragged tiles everywhere (odd extents through every stride-2 stage, the 32-pixel minimum)"""
    import detr_oracle as O
    from odam_amd import detector, weights
    sd = weights.make_state_dict(seed=3)
    det = detector.Detector(max_batch=B, device=DEV, n_streams=1)
    det.load_state_dict(sd)
    torch.manual_seed(H * 7 + W)
    img = torch.randn(B, 3, H, W)
    ref = O.detr_forward(sd, img)
    got = det(img.to(DEV))
    for k in KEYS:
        assert (got[k].cpu() - ref[k]).abs().max().item() <= 2e-4 * max(1.0, ref[k].abs().max().item()), k
    assert torch.equal(got["pred_logits"].cpu().argmax(-1), ref["pred_logits"].argmax(-1))
    det.close()


def test_missing_weight_is_an_error():
    from odam_amd import detector, weights, _lib
    sd = weights.make_state_dict(seed=0)
    del sd["transformer.decoder.norm.weight"]
    det = detector.Detector(max_batch=1, device=DEV)
    det.load_state_dict(sd)
    with pytest.raises(_lib.OdamError, match="transformer.decoder.norm.weight"):
        det(torch.zeros(1, 3, 64, 64, device=DEV))


def test_streamed_detection_equals_plain_forward(model):
    """multi-stream, chunked detect_resident_chunks returns the same rows as forward + postprocess_rows"""
    det, _ = model
    torch.manual_seed(11)
    frames = torch.randn(10, 3, 256, 320, device=DEV)
    plain = det.postprocess_rows(det(frames), (640, 480), K)
    got = np.zeros_like(plain)
    seen = []
    for c0, rows in det.detect_resident_chunks(frames, (640, 480), K, chunk=4):
        got[c0:c0 + rows.shape[0]] = rows
        seen.append((c0, rows.shape[0]))
    assert sum(n for _, n in seen) == 10 and [c for c, _ in seen] == sorted(c for c, _ in seen)
    assert np.allclose(got, plain, rtol=0, atol=2e-6)
    assert np.array_equal(got[..., 1], plain[..., 1])          # class labels
    assert np.allclose(det.detect_resident(frames, (640, 480), K), plain, rtol=0, atol=2e-6)


@pytest.mark.parametrize("w,h,resize", [(640, 480, (800, 1333)), (800, 600, (800, 1333)), (96, 72, (240, 400)),
                                        (57, 131, (120, 200)), (400, 300, (150, 1333)), (321, 240, (240, 1333))])
def test_device_transform_equals_host_transform(model, w, h, resize):
    """A1 on the device (PIL bilinear resize in 22-bit fixed point, /255, normalise) is bit-identical to the PIL +
    torch host transform, for up-scaling, down-scaling and a pass-through axis"""
    from odam_amd import transforms
    det, _ = model
    rng = np.random.default_rng(w * 1000 + h)
    imgs = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    imgs[1] = (np.add.outer(np.arange(h), np.arange(w))[..., None] * np.array([1, 2, 3])) % 256   # ramps
    imgs[2, ::2] = 255; imgs[2, 1::2] = 0                                                          # extremes
    host = torch.stack([transforms.Transforms(*resize)(im)[0] for im in imgs])
    det.resize = resize
    try:
        got = det.preprocess_u8(torch.from_numpy(imgs).to(DEV)).cpu()
    finally:
        det.resize = (800, 1333)
    assert got.shape == host.shape
    assert torch.equal(got, host)


def test_streamed_detection_from_raw_frames(model):
    """uint8 frames (resident, or pinned on the host and uploaded under the kernels) through the device transform
    give exactly the rows of the host-transformed float frames"""
    from odam_amd import transforms
    det, _ = model
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (6, 72, 96, 3), dtype=np.uint8)
    det.resize = (240, 400)
    try:
        host = torch.stack([transforms.Transforms(240, 400)(im)[0] for im in raw]).to(DEV)
        want = det.detect_resident(host, (96, 72), K)
        got_dev = det.detect_resident(torch.from_numpy(raw).to(DEV), (96, 72), K)
        got_pin = det.detect_resident(torch.from_numpy(raw).pin_memory(), (96, 72), K)
    finally:
        det.resize = (800, 1333)
    assert np.array_equal(got_dev, want) and np.array_equal(got_pin, want)


def test_detect_frames_with_device_transforms(model):
    """OdamProcess.detect_frames: raw frames + DeviceTransforms give the rows of the host-transform path"""
    from PIL import Image
    from odam_amd import synth, transforms
    from odam_amd.processor import OdamProcess
    det, _ = model
    frames = [Image.fromarray(f) for f in synth.make_frames(3, h=96, w=128, seed=2)]
    rows = []
    for t in (transforms.Transforms(192, 320), transforms.DeviceTransforms(192, 320)):
        proc = OdamProcess(det, None, t, None, detect_threshold=0.02)
        proc.init_sequence(K, 96, 128)
        rows.append(proc.detect_frames(frames, [7, 8, 9]))
    det.resize = (800, 1333)
    assert sum(len(r) for r in rows[0]) > 0
    assert [len(r) for r in rows[0]] == [len(r) for r in rows[1]]
    for a, b in zip(rows[0], rows[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


# ---- bf16 mode (BASELINE config 4): bf16 weights/activations, bf16 MFMA, fp32 accumulation ------------------------
def _to_bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,relu,res", [
    (2, 25, 34, 256, 512, 1, 1, 0, True, True),
    (1, 40, 53, 64, 64, 3, 1, 1, True, False),
    (2, 33, 41, 128, 128, 3, 2, 1, False, False),
    (1, 64, 80, 3, 64, 7, 2, 3, True, False),       # stem: Cin 3 -> 8, K 392 -> 448
    (1, 30, 40, 512, 384, 1, 1, 0, True, True),     # ragged second 256-column tile
    (1, 20, 27, 64, 256, 3, 1, 1, True, False),     # one 64-channel chunk, nine taps, ragged row tile
    (1, 16, 16, 32, 128, 1, 1, 0, False, False),    # K = 32 padded to 64: fewer k-tiles than ring stages
    (3, 25, 34, 1024, 256, 1, 1, 0, True, False),   # long K (32 k-tiles), ten row tiles
    (2, 26, 35, 256, 256, 3, 2, 1, True, True),     # stride 2, odd size, two images per row tile
])
def test_conv_gemm_bf16_vs_torch(B, H, W, Cin, Cout, k, s, p, relu, res):
    """bf16 operands are exact in fp32, so against a float conv on the bf16-rounded operands only the summation
    order differs (fp32 output requested); the bf16-stored output then agrees to bf16 rounding."""
    from odam_amd import _lib
    L = _lib.lib()
    torch.manual_seed(B * 100 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W).to(torch.bfloat16).float(); w = (torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(torch.bfloat16).float()
    sc = torch.rand(Cout) + 0.5; bi = torch.randn(Cout)
    y = F.conv2d(x, w, None, s, p) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)
    r = torch.randn_like(y).to(torch.bfloat16).float() if res else None
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    k_order = 1 if Cin % 64 == 0 and k > 1 else 0
    wpk, CinP, Kpad = _pack(w, 8, k_order)
    xh = torch.zeros(B, H, W, CinP); xh[..., :Cin] = x.permute(0, 2, 3, 1)
    dx, dw = _to_bf16_bits(xh).to(DEV), _to_bf16_bits(wpk).to(DEV)
    dsc, dbi = sc.to(DEV), bi.to(DEV)
    dr = _to_bf16_bits(r.permute(0, 2, 3, 1).contiguous()).to(DEV) if res else None
    for out_f32 in (1, 0):
        dy = torch.empty(B, y.shape[2], y.shape[3], Cout, device=DEV, dtype=torch.float32 if out_f32 else torch.bfloat16)
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dsc), _lib.ptr(dbi), _lib.ptr(dr), _lib.ptr(dy),
                                              B, H, W, CinP, Cout, k, k, s, p, Kpad, int(relu), out_f32, k_order, _st()), "conv bf16")
        got = dy.float().cpu().permute(0, 3, 1, 2)
        tol = 3e-6 if out_f32 else 8e-3     # bf16 has 8 significant bits
        assert (got - y).abs().max().item() <= tol * max(y.abs().max().item(), 1.0), out_f32


@pytest.mark.parametrize("B,H,W,P,stride,PN,res", [
    (2, 37, 45, 64, 1, 0, True),       # layer1 shape, ragged last tile (3330 rows)
    (2, 37, 45, 64, 1, 64, True),      # ... + the next block's reduce on the tile
    (1, 40, 52, 64, 1, 128, True),     # layer1's last block chained into layer2's first reduce
    (2, 33, 41, 128, 1, 128, True),
    (2, 33, 41, 128, 2, 0, True),      # a stage's first block: stride 2 on the 3x3
    (3, 25, 34, 256, 1, 0, True),      # layer3 shape
    (2, 26, 35, 256, 2, 0, False),     # stride 2, odd size, no residual
])
def test_bf16_bottleneck_on_the_tile_is_bit_identical(B, H, W, P, stride, PN, res):
    """3x3 + expand + residual (+ next reduce) as ONE launch (conv_gemm.hip fused_bf16) against the same layers launched one by
    one: same products in the same order, same rounding points -> the same bf16 bits"""
    from odam_amd import _lib
    L = _lib.lib()
    torch.manual_seed(B * 100 + P + PN + stride)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = torch.randn(B, H, W, P)
    w2 = torch.randn(P, P, 3, 3) / (9 * P) ** 0.5
    w3 = torch.randn(4 * P, P, 1, 1) / P ** 0.5
    w1n = torch.randn(max(PN, 1), 4 * P, 1, 1) / (4 * P) ** 0.5
    s2, b2 = torch.rand(P) + 0.5, torch.randn(P) * 0.3
    s3, b3 = torch.rand(4 * P) + 0.5, torch.randn(4 * P) * 0.3
    s1, b1 = torch.rand(max(PN, 1)) + 0.5, torch.randn(max(PN, 1)) * 0.3
    r = torch.randn(B, Ho, Wo, 4 * P)
    w2p, _, K2 = _pack(w2, 8, 1); w3p, _, K3 = _pack(w3, 8, 0); w1p, _, K1 = _pack(w1n, 8, 0)
    assert K2 == 9 * P and K3 == P and K1 == 4 * P
    d = lambda t: _to_bf16_bits(t.contiguous()).to(DEV)
    dx, dw2, dw3, dw1, dr = d(x), d(w2p), d(w3p), d(w1p), d(r) if res else None
    f = lambda t: t.contiguous().to(DEV)
    ds2, db2, ds3, db3, ds1, db1 = f(s2), f(b2), f(s3), f(b3), f(s1), f(b1)
    bf = lambda *shape: torch.full(shape, -1, device=DEV, dtype=torch.int16)
    # one by one
    _lib.check(L.odam_op_conv_bf16_mode(2), "mode")
    try:
        t = bf(B, Ho, Wo, P); y0 = bf(B, Ho, Wo, 4 * P); n0 = bf(B, Ho, Wo, max(PN, 1))
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(dx), _lib.ptr(dw2), _lib.ptr(ds2), _lib.ptr(db2), None, _lib.ptr(t), B, H, W, P, P, 3, 3,
                                              stride, 1, K2, 1, 0, 1, _st()), "c2")
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(t), _lib.ptr(dw3), _lib.ptr(ds3), _lib.ptr(db3), _lib.ptr(dr), _lib.ptr(y0), B, Ho, Wo, P,
                                              4 * P, 1, 1, 1, 0, K3, 1, 0, 0, _st()), "c3")
        if PN:
            _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(y0), _lib.ptr(dw1), _lib.ptr(ds1), _lib.ptr(db1), None, _lib.ptr(n0), B, Ho, Wo, 4 * P,
                                                  PN, 1, 1, 1, 0, K1, 1, 0, 0, _st()), "c1n")
        y1 = bf(B, Ho, Wo, 4 * P); n1 = bf(B, Ho, Wo, max(PN, 1))
        fuse = _lib.config()["cg.fuse_bf16"]
        if fuse == 0 or (PN and fuse < 2):
            pytest.skip("the fused bf16 bottleneck is switched off in this run (cg.fuse_bf16)")
        _lib.check(L.odam_op_bottleneck_bf16(_lib.ptr(dx), _lib.ptr(dw2), _lib.ptr(ds2), _lib.ptr(db2), _lib.ptr(dw3), _lib.ptr(ds3), _lib.ptr(db3),
                                             _lib.ptr(dr), _lib.ptr(y1), _lib.ptr(dw1) if PN else None, _lib.ptr(ds1), _lib.ptr(db1),
                                             _lib.ptr(n1) if PN else None, B, H, W, P, stride, PN, _st()), "bottleneck")
        torch.cuda.synchronize()
    finally:
        _lib.check(L.odam_op_conv_bf16_mode(1), "mode")
    assert torch.equal(y1, y0), (y1 != y0).float().mean().item()
    if PN:
        assert torch.equal(n1, n0), (n1 != n0).float().mean().item()
    # and the one-by-one result is the convolution (sanity: against torch on the bf16-rounded operands)
    rb = lambda t_: t_.to(torch.bfloat16).float()
    mid = rb(F.relu(F.conv2d(rb(x).permute(0, 3, 1, 2), rb(w2), None, stride, 1) * s2.view(1, -1, 1, 1) + b2.view(1, -1, 1, 1)))
    out = F.conv2d(mid, rb(w3)) * s3.view(1, -1, 1, 1) + b3.view(1, -1, 1, 1)
    if res:
        out = out + rb(r).permute(0, 3, 1, 2)
    out = F.relu(out)
    got = y1.view(torch.bfloat16).float().cpu().permute(0, 3, 1, 2)
    assert (got - out).abs().max().item() <= 1.6e-2 * max(1.0, out.abs().max().item())


@pytest.mark.parametrize("P,PN,drop", [
    (64, 64, "scale"), (64, 0, "bias"), (128, 128, "both"), (256, 0, "scale"), (256, 0, "both"), (128, 0, "bias"),
])
def test_bf16_bottleneck_on_the_tile_without_scale_or_bias(P, PN, drop):
    """The fused tail reads scale / bias through buffer descriptors that hold NO records for a null pointer (round 6: no branch around a
    load): a missing scale must act as 1, a missing bias as 0, on every layer of the launch -- the same bits as the layers one by one."""
    from odam_amd import _lib
    L = _lib.lib()
    B, H, W = 2, 33, 41
    torch.manual_seed(P + PN + len(drop))
    x = torch.randn(B, H, W, P)
    w2 = torch.randn(P, P, 3, 3) / (9 * P) ** 0.5
    w3 = torch.randn(4 * P, P, 1, 1) / P ** 0.5
    w1n = torch.randn(max(PN, 1), 4 * P, 1, 1) / (4 * P) ** 0.5
    r = torch.randn(B, H, W, 4 * P)
    w2p, _, K2 = _pack(w2, 8, 1); w3p, _, K3 = _pack(w3, 8, 0); w1p, _, K1 = _pack(w1n, 8, 0)
    d = lambda t: _to_bf16_bits(t.contiguous()).to(DEV)
    dx, dw2, dw3, dw1, dr = d(x), d(w2p), d(w3p), d(w1p), d(r)
    f = lambda t: t.contiguous().to(DEV)
    sc = lambda n: None if drop in ("scale", "both") else f(torch.rand(n) + 0.5)
    bi = lambda n: None if drop in ("bias", "both") else f(torch.randn(n) * 0.3)
    ds2, db2, ds3, db3, ds1, db1 = sc(P), bi(P), sc(4 * P), bi(4 * P), sc(max(PN, 1)), bi(max(PN, 1))
    p_ = lambda t: _lib.ptr(t) if t is not None else None
    bf = lambda *shape: torch.full(shape, -1, device=DEV, dtype=torch.int16)
    _lib.check(L.odam_op_conv_bf16_mode(2), "mode")
    try:
        t = bf(B, H, W, P); y0 = bf(B, H, W, 4 * P); n0 = bf(B, H, W, max(PN, 1))
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(dx), _lib.ptr(dw2), p_(ds2), p_(db2), None, _lib.ptr(t), B, H, W, P, P, 3, 3, 1, 1, K2, 1, 0, 1, _st()), "c2")
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(t), _lib.ptr(dw3), p_(ds3), p_(db3), _lib.ptr(dr), _lib.ptr(y0), B, H, W, P, 4 * P, 1, 1, 1, 0, K3, 1, 0, 0,
                                              _st()), "c3")
        if PN:
            _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(y0), _lib.ptr(dw1), p_(ds1), p_(db1), None, _lib.ptr(n0), B, H, W, 4 * P, PN, 1, 1, 1, 0, K1, 1, 0, 0,
                                                  _st()), "c1n")
        y1 = bf(B, H, W, 4 * P); n1 = bf(B, H, W, max(PN, 1))
        fuse = _lib.config()["cg.fuse_bf16"]
        if fuse == 0 or (PN and fuse < 2):
            pytest.skip("the fused bf16 bottleneck is switched off in this run (cg.fuse_bf16)")
        _lib.check(L.odam_op_bottleneck_bf16(_lib.ptr(dx), _lib.ptr(dw2), p_(ds2), p_(db2), _lib.ptr(dw3), p_(ds3), p_(db3), _lib.ptr(dr), _lib.ptr(y1),
                                             _lib.ptr(dw1) if PN else None, p_(ds1), p_(db1), _lib.ptr(n1) if PN else None, B, H, W, P, 1, PN, _st()), "bottleneck")
        torch.cuda.synchronize()
    finally:
        _lib.check(L.odam_op_conv_bf16_mode(1), "mode")
    assert torch.equal(y1, y0), (y1 != y0).float().mean().item()
    if PN:
        assert torch.equal(n1, n0), (n1 != n0).float().mean().item()
    # and it is the convolution without the dropped terms (against torch on the bf16-rounded operands)
    rb = lambda t_: t_.to(torch.bfloat16).float()
    one = lambda v, n: torch.ones(n) if v is None else v.cpu()
    zero = lambda v, n: torch.zeros(n) if v is None else v.cpu()
    mid = rb(F.relu(F.conv2d(rb(x).permute(0, 3, 1, 2), rb(w2), None, 1, 1) * one(ds2, P).view(1, -1, 1, 1) + zero(db2, P).view(1, -1, 1, 1)))
    out = F.relu(F.conv2d(mid, rb(w3)) * one(ds3, 4 * P).view(1, -1, 1, 1) + zero(db3, 4 * P).view(1, -1, 1, 1) + rb(r).permute(0, 3, 1, 2))
    got = y1.view(torch.bfloat16).float().cpu().permute(0, 3, 1, 2)
    assert (got - out).abs().max().item() <= 1.6e-2 * max(1.0, out.abs().max().item())


def _bf16_check(out, ref_b, ref_f, measured, tag, ref_b2=None):
    """A whole bf16 forward against (a) the fp32 oracle and (b) the bf16-faithful restatement
    (oracle/detr_oracle.py::detr_forward_bf16, which rounds where the kernels round).  Rounding is chaotic over ~120 layers
    (see test_bf16_layers_teacher_forced, which pins the rounding points layer by layer), so the kernel cannot track (b)
    closer than two bf16 evaluations track each other; what CAN be required, and is:
      * the kernel is no farther from the fp32 truth than the faithful bf16 evaluation is (rms within 1.3x, max within 2x):
        the bf16 mode adds no error beyond bf16 storage itself;
      * kernel vs (b) stays inside that same noise level (max within 2x of (b) vs fp32);
      * class labels equal (b)'s wherever (b)'s decision is clear of the noise (top-2 logit gap above twice the largest
        logit deviation between (b) and fp32), and on at least 93 % of all queries (ties at the noise level included);
      * with ref_b2 = (b) evaluated three more times on inputs nudged by one bf16 ulp in 1 % of the pixels (faithful evaluations that
        round independently): the kernel disagrees with (b) no more than (b) disagrees with ITSELF -- labels within 3 % of the 200
        queries of the mean self-agreement (measured: kernel 0.955; nudged oracle 0.945 ... 0.975, mean 0.958 at 1 % of the pixels,
        0.977 at 0.1 %), every output's rms deviation within 1.5x of the nudged oracle's (measured 0.76 ... 0.91x even against the
        0.1 % nudge: the kernel tracks the oracle more closely than the oracle tracks its nudged self).  tests/golden/c4_margins.txt
        (probe_c4_margins.py) is why the gate is relative: on the best linear class heads there are for this network two faithful
        bf16 evaluations agree on 96-99.5 % of the labels, so a fixed '>= 0.99' would test the weights' margins, not the kernels."""
    def rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12)).item()

    def mx(a, b):
        return ((a - b).abs().max() / b.abs().max().clamp_min(1.0)).item()
    for k in KEYS:
        assert out[k].dtype == torch.float32
        g, b_, f = out[k].cpu(), ref_b[k], ref_f[k]
        for nm, v in (("gpu_vs_fp32_rms", rms(g, f)), ("bf16oracle_vs_fp32_rms", rms(b_, f)), ("gpu_vs_bf16oracle_rms", rms(g, b_)),
                      ("gpu_vs_fp32_max", mx(g, f)), ("bf16oracle_vs_fp32_max", mx(b_, f)), ("gpu_vs_bf16oracle_max", mx(g, b_))):
            measured(f"detr_bf16.{tag}.{k}.{nm}", v)
        assert rms(g, f) <= 1.3 * rms(b_, f) + 1e-4, (k, rms(g, f), rms(b_, f))
        assert mx(g, f) <= 2.0 * mx(b_, f) + 1e-4, (k, mx(g, f), mx(b_, f))
        assert mx(g, b_) <= 2.0 * mx(b_, f) + 1e-4, (k, mx(g, b_), mx(b_, f))
        assert mx(g, f) <= 0.1, (k, mx(g, f))      # sanity cap (bf16 storage through ~120 layers: measured up to 0.09 on the angle head)
    lab, lab_b = out["pred_logits"].cpu().argmax(-1), ref_b["pred_logits"].argmax(-1)
    noise = (ref_b["pred_logits"] - ref_f["pred_logits"]).abs().max().item()
    top2 = ref_b["pred_logits"].topk(2, -1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 2.0 * noise
    measured(f"detr_bf16.{tag}.label_disagreement_share", (lab != lab_b).float().mean().item())
    measured(f"detr_bf16.{tag}.decisive_share", decisive.float().mean().item())
    assert decisive.float().mean().item() >= 0.2, "fixture without decisive logits"      # measured: 1.0 at 192x256, 0.26 at full size
    assert torch.equal(lab[decisive], lab_b[decisive])
    assert (lab == lab_b).float().mean().item() >= 0.93      # all queries, ties at the bf16 noise level included (measured 0.965 ... 1.0)
    if ref_b2 is not None:      # a list of nudged evaluations: one alone is a noisy yardstick (label agreement 0.945 ... 0.975 over nudge seeds)
        agree = [(r["pred_logits"].argmax(-1) == lab_b).float().mean().item() for r in ref_b2]
        self_agree = sum(agree) / len(agree)
        measured(f"detr_bf16.{tag}.label_agreement_oracle_vs_nudged_self", self_agree)
        measured(f"detr_bf16.{tag}.label_agreement_oracle_vs_nudged_self_min", min(agree))
        measured(f"detr_bf16.{tag}.label_agreement_gpu_vs_oracle", (lab == lab_b).float().mean().item())
        assert (lab == lab_b).float().mean().item() >= self_agree - 0.03, ((lab == lab_b).float().mean().item(), agree)
        for k in KEYS:
            r_self = sum(rms(r[k], ref_b[k]) for r in ref_b2) / len(ref_b2)
            r_gpu = rms(out[k].cpu(), ref_b[k])
            measured(f"detr_bf16.{tag}.{k}.oracle_vs_nudged_self_rms", r_self)
            measured(f"detr_bf16.{tag}.{k}.gpu_over_nudged_self_rms", r_gpu / max(r_self, 1e-12))
            assert r_gpu <= 1.5 * r_self + 1e-4, (k, r_gpu, r_self)      # measured: 0.76 ... 0.89 -- the kernel tracks the oracle more closely than the oracle tracks its nudged self


@pytest.mark.parametrize("ring", [1, 2])      # 2: every eligible layer on the 256-row ring kernel (3x3 stride 1: the window main loop)
def test_bf16_layers_teacher_forced(measured, ring):
    """Every contraction of config 4's network (R101 + transformer + heads: 104 convolutions, 86 linear layers) on the
    bf16-faithful oracle's OWN inputs: the kernel must return the oracle's bf16 bits except for one-ulp ties of the fp32
    summation order (a sum that lands within ~1e-6 of a bf16 rounding boundary).  This is what pins the rounding points of the
    bf16 mode -- a whole forward cannot: two evaluations that differ by a fraction of a bf16 ulp round independently from
    there on, so end to end the kernel and ANY restatement decorrelate to the bf16 noise level (measured below)."""
    import detr_oracle as O
    from odam_amd import _lib, weights
    L = _lib.lib()
    sd = weights.make_state_dict(backbone="resnet101", seed=0, scene=True)
    torch.manual_seed(3)
    img = torch.randn(1, 3, 192, 256)
    O.TRACE = []
    try:
        O.detr_forward_bf16(sd, img, blocks=(3, 4, 23, 3))
        trace = O.TRACE
    finally:
        O.TRACE = None
    assert len(trace) >= 104 + 80
    worst_frac, worst_ulp, n_checked = 0.0, 0, 0
    _lib.check(L.odam_op_conv_bf16_mode(ring), "mode")
    for rec in trace:
        x, w, y = rec["x"], rec["w"], rec["y"]
        if rec["kind"] == "linear":                      # [B, M, K] rows -> a 1 x (B M) image with K channels
            K_ = x.shape[-1]
            x = x.reshape(1, -1, K_).permute(0, 2, 1).unsqueeze(2)          # [1, K, 1, BM]
            w = w.reshape(w.shape[0], K_, 1, 1)
            y = y.reshape(1, -1, y.shape[-1]).permute(0, 2, 1).unsqueeze(2)
            res = rec["res"].reshape(1, -1, y.shape[1]).permute(0, 2, 1).unsqueeze(2) if rec["res"] is not None else None
        else:
            res = rec["res"]
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[2]
        k_order = 1 if Cin % 64 == 0 and k > 1 else 0
        wpk, CinP, Kpad = _pack(w, 8, k_order)
        xh = torch.zeros(B, H, W, CinP); xh[..., :Cin] = x.permute(0, 2, 3, 1)
        sc = rec["scale"] if rec["scale"] is not None else None
        dx, dw = _to_bf16_bits(xh).to(DEV), _to_bf16_bits(wpk).to(DEV)
        dsc = sc.contiguous().to(DEV) if sc is not None else None
        dbi = rec["bias"].contiguous().to(DEV) if rec["bias"] is not None else None
        dr = _to_bf16_bits(res.permute(0, 2, 3, 1).contiguous()).to(DEV) if res is not None else None
        Ho, Wo = y.shape[2], y.shape[3]
        f32 = bool(rec["out_f32"])
        dy = torch.empty(B, Ho, Wo, Cout, device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
        _lib.check(L.odam_op_conv2d_nhwc_bf16(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dsc), _lib.ptr(dbi), _lib.ptr(dr), _lib.ptr(dy),
                                              B, H, W, CinP, Cout, k, k, rec["stride"], rec["padding"], Kpad, int(rec["relu"]),
                                              int(f32), k_order, _st()), "conv bf16")
        want = y.permute(0, 2, 3, 1).contiguous()
        if f32:       # the head outputs: fp32, summation order only
            err = (dy.cpu() - want).abs().max().item() / max(1.0, want.abs().max().item())
            measured("detr_bf16.teacher_forced.f32_out_rel", err)
            assert err <= 1e-5, (rec["name"], err)
            continue
        gb, wb = dy.cpu().view(torch.int16).to(torch.int32), _to_bf16_bits(want).to(torch.int32)
        diff = (gb - wb).abs()
        # +0 / -0 (ReLU of a tiny negative vs a tiny positive sum) count as equal
        zero = ((gb & 0x7fff) == 0) & ((wb & 0x7fff) == 0)
        diff = torch.where(zero, torch.zeros_like(diff), diff)
        # near a cancellation (|y| far below the layer's scale) the fp32 summation noise is an ABSOLUTE quantity -- a few 1e-7 of
        # the terms -- and can span several ulps of a tiny result: such elements must agree to 1e-5 of the layer's range
        small = (dy.cpu().float() - want).abs() <= 1e-5 * want.abs().max().item()
        bad = (diff > 1) & ~small
        frac, ulp = (diff != 0).float().mean().item(), int(torch.where(small, diff.clamp(max=1), diff).max().item())
        worst_frac, worst_ulp = max(worst_frac, frac), max(worst_ulp, ulp)
        n_checked += 1
        assert not bad.any() and frac <= 5e-3, (rec["name"], rec["kind"], tuple(x.shape), frac, int(diff.max().item()))
    _lib.check(L.odam_op_conv_bf16_mode(1), "mode")
    measured("detr_bf16.teacher_forced.worst_mismatch_share", worst_frac)
    measured("detr_bf16.teacher_forced.worst_ulp", worst_ulp)
    assert n_checked >= 170


def test_bf16_forward_vs_bf16_faithful_oracle(measured):
    """R101 bf16 (config 4's network) at a reduced size, scene weights (spread logits, 8 live classes)"""
    import detr_oracle as O
    from odam_amd import detector, weights
    sd = weights.make_state_dict(backbone="resnet101", seed=0, scene=True)
    det = detector.Detector(backbone="resnet101", max_batch=2, device=DEV, dtype="bf16", n_streams=1)
    det.load_state_dict(sd)
    torch.manual_seed(3)
    img = torch.randn(2, 3, 192, 256)
    ref_b = O.detr_forward_bf16(sd, img, blocks=(3, 4, 23, 3))
    ref_f = O.detr_forward(sd, img, blocks=(3, 4, 23, 3))
    out = det(img.to(DEV))
    _bf16_check(out, ref_b, ref_f, measured, "small")
    det.close()


def test_bf16_r101_full_size_vs_bf16_faithful_oracle(measured):
    """BASELINE config 4 at its full size: DETR-R101, 800x600 frames resized to 3x800x1066, bf16 weights and activations on
    the bf16 matrix instruction, batch of 2 (the ring kernel and every large-layer path apply)"""
    import detr_oracle as O
    from odam_amd import detector, synth, weights
    sd = weights.make_state_dict(backbone="resnet101", seed=0, scene=True)
    det = detector.Detector(backbone="resnet101", max_batch=2, device=DEV, dtype="bf16", n_streams=1)
    det.load_state_dict(sd)
    raw = np.stack(list(synth.make_frames(2, h=600, w=800, seed=4)))
    img = det.preprocess_u8(torch.from_numpy(raw).to(DEV))
    assert tuple(img.shape) == (2, 3, 800, 1066)
    torch.set_num_threads(8)
    ref_b = O.detr_forward_bf16(sd, img.cpu(), blocks=(3, 4, 23, 3))
    ref_f = O.detr_forward(sd, img.cpu(), blocks=(3, 4, 23, 3))
    ref_b2 = []      # three more, independently rounding evaluations of the oracle: one bf16 ulp on 1 % of the pixels (the kernel's own
    for seed in (1, 2, 3):      # deviations -- one-ulp ties of the fp32 summation order -- start in every layer, in 3e-4 ... 5e-3 of the elements)
        nudge = torch.rand(img.shape, generator=torch.Generator().manual_seed(seed)) < 1e-2
        ref_b2.append(O.detr_forward_bf16(sd, torch.where(nudge, img.cpu() * (1 + 2.0 ** -7), img.cpu()), blocks=(3, 4, 23, 3)))
    out = det(img)
    _bf16_check(out, ref_b, ref_f, measured, "full", ref_b2)
    det.close()


@pytest.mark.parametrize("cfg", ["cg.ut=0", "cg.tiles=0", "cg.tiles=7", "cg.tiles=15", "cg.force=2", "cg.ring=2", "cg.ring=0", "cg.ring=2,cg.presplit=0", "cg.f32=0",
                                 "att.x3=0,cg.fuse=1", "stem.rows=0,cg.fuse=0", "stem.pool=0", "cg.s1=0", "cg.fuse_bf16=0", "cg.fuse_bf16=1,att.bf16_mfma=0", "cg.mfma16=0", "cg.small_x3=0"])
def test_conv_gemm_alternative_paths(cfg):
    """Every surviving switch of the library's config table (include/odam_config.h): the register-staged gather (what layers too
    large for 31-bit buffer offsets fall back to), the 4-wave tiles, the four-stage small-tile pipeline, the ring kernel
    forced onto / kept off every eligible shape, fp32 split without pre-split filters, the fp32 matrix instruction everywhere,
    attention on the fp32 instruction, the small tiles on the fp32 instruction, bottlenecks as separate launches (fp32 and bf16), conv1 on the small tiles, the bf16 3x3
    through the generic tap gather: same parity.  The nested run's host side (odam_amd/_lib.py) pushes ODAM_CONFIG through
    odam_config_set when it loads the library."""
    import os
    import subprocess
    import sys
    if os.environ.get("ODAM_NESTED"):
        pytest.skip("already inside the nested run")
    sel = "conv_gemm_vs_torch or conv_gemm_bf16"
    if cfg in ("cg.f32=0", "cg.mfma16=0", "cg.small_x3=0"):      # the reference-run goldens and the oracle comparisons must hold on the fp32 instruction as well
        sel += " or reference_golden or vs_oracle or mixed_size"
    if "stem." in cfg or "cg.fuse=" in cfg:
        sel = "reference_golden or vs_oracle"
    if "att.x3" in cfg:
        sel = "attention_vs_torch or reference_golden or vs_oracle or mixed_size"
    if "cg.s1" in cfg or "cg.fuse_bf16" in cfg:
        sel = "bf16"
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-m", "gpu", "-q", "-x", "-k", sel],
                       env={**os.environ, "ODAM_CONFIG": cfg, "ODAM_NESTED": "1"}, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_pooled_stem_is_bit_identical():
    """conv1 with the 3x3 / stride 2 max-pool on its tile (odam_config stem.pool, ConvGemmArgs::pool: 2-D patches of conv outputs
    pooled in LDS, the 1.75 GB stem tensor of a 32-frame batch never written) against conv1 + the max-pool kernel: the same
    products in the same order and an exact maximum, so every output of the forward is equal bit for bit -- at full size (3 frames:
    patches at all four image borders, a ragged last patch column; fp32 and bf16) and on a small odd-sized input through the pinned
    kernel choice."""
    from odam_amd import _lib, detector, weights
    sd = weights.make_state_dict(seed=0, scene=True)
    try:
        for shape, pin, dtype in (((3, 3, 800, 1066), 0, "fp32"), ((2, 3, 230, 310), 1, "fp32"), ((3, 3, 800, 1066), 0, "bf16")):
            _lib.set_config("cg.pin", pin)
            det = detector.Detector(max_batch=shape[0], device=DEV, n_streams=1, dtype=dtype)
            det.load_state_dict(sd)
            torch.manual_seed(3)
            img = torch.randn(*shape, device=DEV)
            outs = []
            L = _lib.lib()
            L.odam_op_pooled_stem_launches.restype = ctypes.c_longlong
            for v in (1, 0):
                _lib.set_config("stem.pool", v)
                n0 = L.odam_op_pooled_stem_launches()
                outs.append({k: t.clone() for k, t in det(img).items() if torch.is_tensor(t)})
                assert L.odam_op_pooled_stem_launches() - n0 == v      # the path under test is the one that ran
            det.close()
            for k in KEYS:
                assert torch.equal(outs[0][k], outs[1][k]), (shape, dtype, k)
    finally:
        _lib.set_config("stem.pool", 1); _lib.set_config("cg.pin", 0)


def test_mixed_size_batch_vs_reference(golden):
    """Three images of different sizes in one call: the reference's NestedTensor path (zero padding to the batch
    maximum, mask -> token grid, per-image sine embedding, key_padding_mask in encoder self- and decoder
    cross-attention; misc.py:303-320, backbone.py:79, transformer.py:157-160) against its own run (detr_mixed.npz)."""
    from odam_amd import detector, weights
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden_detr_mixed import images
    z = golden("detr_mixed.npz")
    det = detector.Detector(max_batch=4, device=DEV, n_streams=1)
    det.load_state_dict(weights.make_state_dict(seed=0, scene=True))
    out = det(images())                              # a list of different shapes -> forward_nested
    for k in KEYS:
        if k in z.files:
            err = (out[k].cpu() - torch.from_numpy(z[k])).abs().max().item()
            assert err <= 5e-4 * max(1.0, np.abs(z[k]).max()), (k, err)
    assert torch.equal(out["pred_logits"].cpu().argmax(-1), torch.from_numpy(z["pred_logits"]).argmax(-1))
    # the padding is not a no-op: image 0 alone gives different logits, and we reproduce those too
    alone = det(images()[0][None])["pred_logits"].cpu().numpy()
    assert np.abs(alone - z["alone_logits0"]).max() <= 5e-4 * np.abs(z["alone_logits0"]).max()
    assert np.abs(z["pred_logits"][0] - z["alone_logits0"][0]).max() > 0.1
    det.close()
