"""Sinkhorn launch alone: 100 iterations at the association's sizes, probability-domain wavefront against the log-space loop
(odam_config assoc.sk_fast 1 / 0); HIP event pair around 200 launches"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import _lib
dev = "cuda:0"
L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(torch.device(dev)).cuda_stream)
for m, n in ((10, 8), (40, 12), (42, 21), (60, 30), (100, 21), (127, 30), (200, 30)):
    sc = torch.randn(m, n, device=dev) * 5
    out = torch.empty(m + 1, n + 1, device=dev)
    line = f"{m:4d} x {n:2d}:"
    for fast in (0, 1):
        _lib.set_config("assoc.sk_fast", fast)
        for _ in range(5): L.odam_assoc_sinkhorn(_lib.ptr(sc), n, m, n, ctypes.c_float(1.0), 100, _lib.ptr(out), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): L.odam_assoc_sinkhorn(_lib.ptr(sc), n, m, n, ctypes.c_float(1.0), 100, _lib.ptr(out), st)
        e1.record(); torch.cuda.synchronize()
        line += f"  sk_fast={fast}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us"
    print(line, flush=True)
_lib.set_config("assoc.sk_fast", 1)
