import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.nn.functional as F
from odam_amd import _lib, weights, detector
import detr_oracle as O
L = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
torch.manual_seed(0)

def pack(w):  # [Cout,Cin,KH,KW] -> [Cout][Kpad]
    Cout, Cin, KH, KW = w.shape
    CinP = (Cin + 3) // 4 * 4
    wp = torch.zeros(Cout, KH, KW, CinP); wp[..., :Cin] = w.permute(0, 2, 3, 1)
    K = KH * KW * CinP; Kpad = (K + 31) // 32 * 32
    out = torch.zeros(Cout, Kpad); out[:, :K] = wp.reshape(Cout, K)
    return out, CinP, Kpad

def conv_case(B, H, W, Cin, Cout, k, s, p, relu, res, bn):
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    sc = torch.rand(Cout) + 0.5 if bn else None; bi = torch.randn(Cout) if bn else None
    y = F.conv2d(x, w, None, s, p)
    if bn: y = y * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)
    r = torch.randn_like(y) if res else None
    if res: y = y + r
    if relu: y = F.relu(y)
    wp, CinP, Kpad = pack(w)
    xh = torch.zeros(B, H, W, CinP); xh[..., :Cin] = x.permute(0, 2, 3, 1)
    Ho, Wo = y.shape[2:]
    d = lambda t: t.to(dev).contiguous() if t is not None else None
    dx, dw, dsc, dbi = d(xh), d(wp), d(sc), d(bi)
    dr = d(r.permute(0, 2, 3, 1)) if res else None
    dy = torch.empty(B, Ho, Wo, Cout, device=dev)
    _lib.check(L.odam_op_conv2d_nhwc(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(dsc), _lib.ptr(dbi), _lib.ptr(dr), _lib.ptr(dy),
               B, H, W, CinP, Cout, k, k, s, p, Kpad, int(relu), 0, st()), "conv")
    got = dy.cpu().permute(0, 3, 1, 2)
    err = (got - y).abs().max().item() / max(y.abs().max().item(), 1e-6)
    print(f"conv B{B} {H}x{W} {Cin}->{Cout} k{k}s{s}p{p} relu={relu} res={res} bn={bn}: rel err {err:.2e}")
    return err

errs = [conv_case(2, 37, 45, 64, 64, 1, 1, 0, True, False, True), conv_case(1, 40, 53, 64, 64, 3, 1, 1, True, False, True),
        conv_case(2, 33, 41, 128, 128, 3, 2, 1, True, False, True), conv_case(1, 80, 106, 3, 64, 7, 2, 3, True, False, True),
        conv_case(2, 25, 34, 256, 1024, 1, 1, 0, True, True, True), conv_case(2, 50, 67, 512, 1024, 1, 2, 0, False, False, True),
        conv_case(1, 1, 300, 256, 19, 1, 1, 0, False, False, True), conv_case(3, 25, 34, 2048, 256, 1, 1, 0, False, False, True)]
assert max(errs) < 2e-5, errs

# attention
for (B, Lq, Lk) in [(2, 850, 850), (3, 100, 100), (2, 100, 850), (1, 33, 70)]:
    q = torch.randn(B, Lq, 256); k = torch.randn(B, Lk, 256); v = torch.randn(B, Lk, 256)
    qh = q.view(B, Lq, 8, 32).transpose(1, 2) * (1.0 / 32) ** 0.5
    kh = k.view(B, Lk, 8, 32).transpose(1, 2); vh = v.view(B, Lk, 8, 32).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).transpose(1, 2).reshape(B, Lq, 256)
    dq, dk, dv = q.to(dev), k.to(dev), v.to(dev); do = torch.empty(B, Lq, 256, device=dev)
    _lib.check(L.odam_op_attention(_lib.ptr(dq), 256, _lib.ptr(dk), 256, _lib.ptr(dv), 256, _lib.ptr(do), 256, B, 8, Lq, Lk, st()), "attn")
    e = (do.cpu() - ref).abs().max().item()
    print(f"attention B{B} Lq{Lq} Lk{Lk}: max abs err {e:.2e}")
    assert e < 2e-5
# layernorm
x = torch.randn(777, 256); r = torch.randn(777, 256); g = torch.rand(256) + 0.5; b = torch.randn(256)
ref = F.layer_norm(x + r, (256,), g, b, 1e-5)
dy = torch.empty(777, 256, device=dev)
_lib.check(L.odam_op_add_layernorm(_lib.ptr(x.to(dev)), _lib.ptr(r.to(dev)), _lib.ptr(g.to(dev)), _lib.ptr(b.to(dev)), _lib.ptr(dy), 777, st()), "ln")
print("layernorm max abs err", (dy.cpu() - ref).abs().max().item())
# maxpool
x = torch.randn(2, 64, 41, 53)
ref = F.max_pool2d(x, 3, 2, 1)
dy = torch.empty(2, ref.shape[2], ref.shape[3], 64, device=dev)
_lib.check(L.odam_op_maxpool3x3s2_nhwc(_lib.ptr(x.permute(0, 2, 3, 1).contiguous().to(dev)), _lib.ptr(dy), 2, 41, 53, 64, st()), "pool")
print("maxpool exact:", torch.equal(dy.cpu().permute(0, 3, 1, 2), ref))

# full model
sd = weights.make_state_dict()
det = detector.Detector(max_batch=8); det.load_state_dict(sd)
for (B, H, W) in [(2, 256, 320), (2, 800, 1066)]:
    img = torch.randn(B, 3, H, W)
    t0 = time.time(); ref = O.detr_forward(sd, img, return_taps=True); tcpu = time.time() - t0
    out = det(img.to(dev)); torch.cuda.synchronize()
    l4, mem = det.debug_taps(B, H, W)
    print(f"--- model B{B} {H}x{W}: oracle CPU {tcpu:.2f}s ({torch.get_num_threads()} threads)")
    def rel(a, b): return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()
    print("  layer4 rel err", rel(l4.cpu(), ref["_layer4"]), "| memory rel err", rel(mem.cpu(), ref["_memory"]), "| layer4 absmax", ref["_layer4"].abs().max().item())
    for k in ("pred_logits", "pred_boxes", "pred_angle", "pred_offset", "pred_size", "pred_depth", "pred_obj_features"):
        print(f"  {k:18s} max abs err {(out[k].cpu() - ref[k]).abs().max().item():.3e}  (ref absmax {ref[k].abs().max().item():.3f})")
    pc = out["pred_logits"].cpu().softmax(-1)[..., :-1]; pr = ref["pred_logits"].softmax(-1)[..., :-1]
    print("  class argmax equal:", torch.equal(pc.argmax(-1), pr.argmax(-1)), "| n above 0.6:", int((pr.max(-1).values > 0.6).sum()),
          "| keep set equal:", torch.equal(pc.max(-1).values > 0.6, pr.max(-1).values > 0.6))
    K = np.array([[577.87, 0, 319.5], [0, 577.87, 239.5], [0, 0, 1]])
    pp = det.postprocess(out, (640, 480), 0.6, K); pref = O.postprocess(ref, (640, 480), 0.6, K)
    for b in range(B):
        same = np.array_equal(pp["classes"][b], pref["classes"][b]) and len(pp["scores"][b]) == len(pref["scores"][b])
        print(f"  frame {b}: {len(pref['scores'][b])} detections; classes equal {same}",
              "| max box diff", float(np.abs(pp["bboxes"][b] - pref["bboxes"][b]).max()) if same and len(pref["scores"][b]) else None,
              "| max t_co diff", float(np.abs(pp["translates"][b] - pref["translates"][b]).max()) if same and len(pref["scores"][b]) else None)
# timing
img = torch.randn(8, 3, 800, 1066, device=dev)
for _ in range(2): det(img)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): det(img)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"forward B=8 800x1066: {dt*1e3:.1f} ms/batch = {8/dt:.1f} frames/s; {163.1*8/dt/1e3:.1f} TFLOP/s")
