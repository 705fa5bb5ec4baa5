import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def run(x, w, y, B, H, W, Cin, Cout, k, s, p, Kpad, n=20):
    for _ in range(3):
        L.odam_op_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), None, None, None, _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 0, 0, st)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        L.odam_op_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), None, None, None, _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (name, B, H, W, Cin, Cout, k, s, p) in [("l3.c2 3x3 256->256", 32, 50, 67, 256, 256, 3, 1, 1), ("l4.c2 3x3 512->512", 32, 25, 34, 512, 512, 3, 1, 1),
                                             ("gemm 16384x2048x2048", 1, 1, 16384, 2048, 2048, 1, 1, 0)]:
    K = k * k * Cin
    y = torch.empty(B * H * W * Cout, device=dev)
    for mode in ("random", "zeros"):
        x = (torch.randn if mode == "random" else torch.zeros)(B, H, W, Cin, device=dev)
        w = (torch.randn if mode == "random" else torch.zeros)(Cout, K, device=dev)
        ms = run(x, w, y, B, H, W, Cin, Cout, k, s, p, K)
        fl = 2.0 * B * H * W * Cout * K
        print(f"{name:24s} {mode:7s}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s", flush=True)
