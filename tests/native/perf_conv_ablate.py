"""Where does conv_gemm's main loop lose time?  Times one layer with parts of the loop compiled out
(-DCG_ABLATE=mask builds in tests/native/_abl/; results are garbage, only the durations mean anything).
usage: python3 tests/native/perf_conv_ablate.py <lib.so> [<lib.so> ...]"""
import os, sys, ctypes
import torch
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(L, x, w, y, B, H, W, Cin, Cout, k, s, p, Kpad, ko, n=20):
    call = lambda: L.odam_op_conv2d_nhwc(P(x), P(w), None, None, None, P(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 0, ko, st)
    for _ in range(3): call()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
layers = [("l1.c2 3x3 64->64", 32, 200, 267, 64, 64, 3, 1, 1), ("l2.c2 3x3 128->128", 32, 100, 134, 128, 128, 3, 1, 1),
          ("l3.c2 3x3 256->256", 32, 50, 67, 256, 256, 3, 1, 1), ("l4.c2 3x3 512->512", 32, 25, 34, 512, 512, 3, 1, 1), ("gemm 16384x2048x2048", 1, 1, 16384, 2048, 2048, 1, 1, 0),
          ("small 64x512x512", 1, 1, 64, 512, 512, 1, 1, 0), ("small 40x256x256", 1, 1, 40, 256, 256, 1, 1, 0),
          ("dec 3200x256x256", 1, 1, 3200, 256, 256, 1, 1, 0), ("dec 3200x2048x256", 1, 1, 3200, 2048, 256, 1, 1, 0),
          ("l1.c1 1x1 64->64", 32, 200, 267, 64, 64, 1, 1, 0), ("l1.c3 1x1 64->256", 32, 200, 267, 64, 256, 1, 1, 0)]
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    for (name, B, H, W, Cin, Cout, k, s, p) in layers:
        K = k * k * Cin
        y = torch.empty(B * H * W * Cout, device=dev)
        x = torch.randn(B, H, W, Cin, device=dev); w = torch.randn(Cout, K, device=dev)
        for ko in ((0, 1) if k > 1 else (0,)):   # random operands: the K order only changes the access pattern
            ms = run(L, x, w, y, B, H, W, Cin, Cout, k, s, p, K, ko)
            fl = 2.0 * B * H * W * Cout * K
            print(f"{os.path.basename(path):16s} {name:24s} k_order {ko}: {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s", flush=True)
