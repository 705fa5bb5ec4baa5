"""WHICH math library does this torch build's CPU path call for the elementwise operations of the super-quadric fit?  (build container only)
Bitwise comparison of torch.sin / cos / exp / log / pow / sigmoid on float32 tensors with (a) Sleef's u10 functions -- the symbols
libtorch_cpu.so itself exports, called through tests/golden/probe_torch_math.c -- and (b) glibc's sinf / cosf / expf / logf / powf.
Result on torch 2.10.0+rocm7.0 (MKL 2024.2, CPU capability AVX512), recorded in probe_torch_math.txt:
  torch.pow(tensor, tensor) == Sleef_powf16_u10 bit for bit; with a Python-scalar exponent the last n % 32 elements come from glibc powf
  torch.sigmoid == 1 / (1 + Sleef_expf16_u10(-x)) bit for bit
  torch.sin / cos / exp / log match NEITHER Sleef nor glibc (1-9 % of the elements differ from each): they are MKL VML calls
  (vsSin / vsCos / vsExp / vsLn are linked into libtorch_cpu.so; ATen's vml.h routes contiguous float tensors to them when MKL is on)
    gcc -O1 -mavx512f -mavx2 -mfma -shared -fPIC -o /tmp/probe_torch_math.so tests/golden/probe_torch_math.c -ldl
    python tests/golden/probe_torch_math.py
"""
import ctypes, numpy as np, torch
S = ctypes.CDLL('/tmp/probe_torch_math.so')
libm = ctypes.CDLL('libm.so.6')
def sleef1(name, x, w=16):
    y = np.empty_like(x); fn = S.sleef1_16 if w == 16 else S.sleef1_8
    rc = fn(name.encode(), x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), len(x)); assert rc == 0, name
    return y
def sleef2(name, x, p, w=16):
    y = np.empty_like(x); fn = S.sleef2_16 if w == 16 else S.sleef2_8
    rc = fn(name.encode(), x.ctypes.data_as(ctypes.c_void_p), p.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), len(x)); assert rc == 0, name
    return y
def glibc1(name, x):
    f = getattr(libm, name); f.restype = ctypes.c_float; f.argtypes = [ctypes.c_float]
    return np.array([f(float(v)) for v in x], np.float32)
def glibc2(name, x, p):
    f = getattr(libm, name); f.restype = ctypes.c_float; f.argtypes = [ctypes.c_float, ctypes.c_float]
    return np.array([f(float(a), float(b)) for a, b in zip(x, p)], np.float32)
rs = np.random.RandomState(0)
def cmp(tag, a, b):
    d = (a.view(np.uint32) != b.view(np.uint32))
    print(f"{tag:50s} differing {d.sum():5d} / {len(a)}   first at {np.nonzero(d)[0][:5]}")
for n in (1000, 1008, 37):
    x = rs.uniform(-3.2, 3.2, n).astype(np.float32)
    t = torch.from_numpy(x)
    print("n =", n)
    cmp("torch.sin vs Sleef_sinf16_u10", torch.sin(t).numpy(), sleef1("Sleef_sinf16_u10", x))
    cmp("torch.sin vs Sleef_sinf8_u10", torch.sin(t).numpy(), sleef1("Sleef_sinf8_u10", x, 8))
    cmp("torch.sin vs glibc sinf", torch.sin(t).numpy(), glibc1("sinf", x))
    cmp("torch.cos vs Sleef_cosf16_u10", torch.cos(t).numpy(), sleef1("Sleef_cosf16_u10", x))
    cmp("torch.cos vs glibc cosf", torch.cos(t).numpy(), glibc1("cosf", x))
    a = np.abs(x) + 1e-3; p = np.full(n, 0.9, np.float32)
    ta = torch.from_numpy(a)
    cmp("torch.pow(t, 0.9) vs Sleef_powf16_u10", torch.pow(ta, 0.9).numpy(), sleef2("Sleef_powf16_u10", a, p))
    cmp("torch.pow(t, 0.9) vs glibc powf", torch.pow(ta, 0.9).numpy(), glibc2("powf", a, p))
    cmp("torch.pow(t, tensor) vs Sleef_powf16_u10", torch.pow(ta, torch.from_numpy(p)).numpy(), sleef2("Sleef_powf16_u10", a, p))
    cmp("torch.exp vs Sleef_expf16_u10", torch.exp(t).numpy(), sleef1("Sleef_expf16_u10", x))
    cmp("torch.exp vs glibc expf", torch.exp(t).numpy(), glibc1("expf", x))
    cmp("torch.log vs Sleef_logf16_u10", torch.log(ta).numpy(), sleef1("Sleef_logf16_u10", a))
    cmp("torch.log vs glibc logf", torch.log(ta).numpy(), glibc1("logf", a))
    cmp("torch.sigmoid vs 1/(1+Sleef exp(-x))", torch.sigmoid(t).numpy(), (np.float32(1) / (np.float32(1) + sleef1("Sleef_expf16_u10", -x))).astype(np.float32))
