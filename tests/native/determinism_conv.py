"""bitwise run-to-run determinism of the contraction kernels, alone and with a second stream keeping the GPU busy"""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib
L = _lib.lib()
SHAPES = [("N64", 8, 200, 267, 256, 64, 1, 1, 0), ("N64-3x3", 8, 200, 267, 64, 64, 3, 1, 1), ("N128-3x3", 8, 100, 134, 128, 128, 3, 1, 1),
          ("N256", 8, 100, 134, 512, 256, 1, 1, 0), ("N256-3x3", 16, 50, 67, 256, 256, 3, 1, 1), ("N1024", 16, 50, 67, 256, 1024, 1, 1, 0),
          ("N512-3x3", 32, 25, 34, 512, 512, 3, 1, 1)]
side = torch.cuda.Stream()
for dt in ("fp32", "bf16"):
    bf = dt == "bf16"; td = torch.bfloat16 if bf else torch.float32
    for name, B, H, W, Cin, Cout, k, s, p in SHAPES:
        kt = 64 if bf else 32
        K = k * k * Cin; Kpad = (K + kt - 1) // kt * kt
        k_order = 1 if (k > 1 and Cin % kt == 0) else 0
        x = torch.randn(B, H, W, Cin, device="cuda").to(td); w = (torch.randn(Cout, Kpad, device="cuda") / K ** 0.5).to(td)
        sc = torch.rand(Cout, device="cuda") + 0.5; bi = torch.randn(Cout, device="cuda")
        ys = [torch.empty(B, H, W, Cout, device="cuda", dtype=td) for _ in range(2)]
        big = torch.randn(8192, 8192, device="cuda")
        def run(y, st):
            sp = ctypes.c_void_p(st.cuda_stream)
            if bf:
                rc = L.odam_op_conv2d_nhwc_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(sc), _lib.ptr(bi), None, _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 1, 0, k_order, sp)
            else:
                rc = L.odam_op_conv2d_nhwc(_lib.ptr(x), _lib.ptr(w), _lib.ptr(sc), _lib.ptr(bi), None, _lib.ptr(y), B, H, W, Cin, Cout, k, k, s, p, Kpad, 1, k_order, sp)
            assert rc == 0
        cur = torch.cuda.current_stream()
        run(ys[0], cur); torch.cuda.synchronize()
        bad = 0; worst = 0.0
        for it in range(30):
            with torch.cuda.stream(side):
                if it % 2: torch.mm(big, big)            # a competing kernel on another stream
                run(ys[1], side) if it % 3 == 0 else None
            run(ys[1], cur); torch.cuda.synchronize()
            if not torch.equal(ys[0].view(torch.int16 if bf else torch.int32), ys[1].view(torch.int16 if bf else torch.int32)):
                bad += 1; worst = max(worst, (ys[0].float() - ys[1].float()).abs().max().item())
        print(f"{dt} {name}: M={B*H*W} N={Cout} K={K}: {bad}/30 runs differ, max |diff| {worst:.3e}", flush=True)
