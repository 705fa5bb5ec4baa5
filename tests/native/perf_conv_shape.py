"""time single bf16 / fp32 layers of the detector at B=32 through the conv op (tile variants via ODAM_CG_FORCE / ODAM_CG_BIG):
   python perf_conv_shape.py [bf16|fp32]"""
import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib
L = _lib.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
bf = dt == "bf16"
es = 2 if bf else 4
td = torch.bfloat16 if bf else torch.float32
SHAPES = [("l1.c3", 32, 200, 267, 64, 256, 1, 1, 0, True), ("l1.c1", 32, 200, 267, 256, 64, 1, 1, 0, False),
          ("l2.c3", 32, 100, 134, 128, 512, 1, 1, 0, True), ("l2.c2", 32, 100, 134, 128, 128, 3, 1, 1, False),
          ("l3.c3", 32, 50, 67, 256, 1024, 1, 1, 0, True), ("l3.c1", 32, 50, 67, 1024, 256, 1, 1, 0, False),
          ("l3.c2", 32, 50, 67, 256, 256, 3, 1, 1, False), ("l4.c2", 32, 25, 34, 512, 512, 3, 1, 1, False)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, H, W, Cin, Cout, k, s, p, res in SHAPES:
    kt = 64 if bf else 32
    K = k * k * Cin
    Kpad = (K + kt - 1) // kt * kt
    k_order = 1 if (k > 1 and Cin % kt == 0) else 0
    x = torch.randn(B, H, W, Cin, device="cuda").to(td)
    w = (torch.randn(Cout, Kpad, device="cuda") / K ** 0.5).to(td)
    sc = torch.rand(Cout, device="cuda") + 0.5; bi = torch.randn(Cout, device="cuda")
    r = torch.randn(B, H, W, Cout, device="cuda").to(td) if res else None
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=td)
    def run():
        if bf:
            rc = L.odam_op_conv2d_nhwc_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(sc), _lib.ptr(bi), _lib.ptr(r), _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 1, 0, k_order, st)
        else:
            rc = L.odam_op_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(sc), _lib.ptr(bi), _lib.ptr(r), _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 1, k_order, st)
        assert rc == 0, L.odam_last_error()
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): run()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / n * 1e6
    M = B * H * W
    fl = 2.0 * M * Cout * K
    by = (M * Cin * (1 if k == 1 else 1) + M * Cout * (2 if res else 1)) * es
    print(f"{name} M={M} N={Cout} K={K}: {us:8.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e6:6.2f} TB/s (compulsory bytes)", flush=True)
