"""(perf_attention_bf16.py [fp32]: with the argument, the fp32 kernel of the headline path -- odam_op_attention, the exact three-way split --
instead.)  The bf16 attention kernel alone (config 4): us per launch and TFLOP/s of QK^T + PV at DETR's shapes, B = 32, 8 heads of 32:
   encoder self-attention 850 x 850, decoder cross-attention 100 x 850, decoder self-attention 100 x 100.   python3 tests/native/perf_attention_bf16.py"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import _lib
L = _lib.lib()
dev = "cuda:0"
FP32 = len(sys.argv) > 1 and sys.argv[1] == "fp32"
DT = torch.float32 if FP32 else torch.bfloat16
B, H = 32, 8
torch.manual_seed(0)
for Lq, Lk in ((850, 850), (100, 850), (100, 100)):
    q = torch.randn(B * Lq, 256, device=dev).to(DT); k = torch.randn(B * Lk, 256, device=dev).to(DT)
    v = torch.randn(B * Lk, 256, device=dev).to(DT); o = torch.empty(B * Lq, 256, device=dev, dtype=DT)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: _lib.check((L.odam_op_attention if FP32 else L.odam_op_attention_bf16)(_lib.ptr(q), 256, _lib.ptr(k), 256, _lib.ptr(v), 256, _lib.ptr(o), 256, B, H, Lq, Lk, st), "att")
    for _ in range(5): run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = sorted(ts)[len(ts) // 2]
    fl = 4.0 * B * H * Lq * Lk * 32
    ref = torch.softmax((q.float().view(B, Lq, H, 32).transpose(1, 2) @ k.float().view(B, Lk, H, 32).transpose(1, 2).transpose(2, 3)) * 32 ** -0.5, -1) @ v.float().view(B, Lk, H, 32).transpose(1, 2)
    err = (o.float().view(B, Lq, H, 32).transpose(1, 2) - ref).abs().max().item()
    import hashlib
    print(f"Lq {Lq:4d} Lk {Lk:4d}: {us:7.1f} us per launch, {fl / us / 1e6:6.1f} TFLOP/s = {fl / us / 1e6 / (2500 / 6 if FP32 else 2500):.3f} of {'2.5 PF / 6' if FP32 else '2.5 PF'}; max |o - fp32 reference| {err:.3e}; sha1 {hashlib.sha1(o.cpu().view(torch.int32 if FP32 else torch.int16).numpy().tobytes()).hexdigest()[:12]}")
