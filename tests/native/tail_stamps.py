"""Phase timeline of the layer1 bottleneck launch's tail (fused_chain), diagnostic build only:
   build the library with -DCG_STAMP=1 first (hipcc ... -DCG_STAMP=1 -o odam_amd/libodam_amd.so odam_amd/csrc/*.hip): every stamp
   executes and the library exports odam_cg_tail_stamps; the product build has neither"""
import os, sys, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from odam_amd import detector, weights, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
det = detector.Detector(max_batch=B, n_streams=1); det.load_state_dict(weights.make_state_dict())
img = torch.randn(B, 3, 800, 1066, device="cuda:0")
det(img); torch.cuda.synchronize()
L = _lib.lib()
out = (ctypes.c_ulonglong * 8)()
L.odam_cg_tail_stamps(out, 1)
det(img); torch.cuda.synchronize()
L.odam_cg_tail_stamps(out, 0)
v = list(out); n = max(v[5], 1)
names = ["wait + barrier (x4)", "expand products + slab write (x4)", "row pass: scale, residual, store (x4)", "split of the final values (x4)",
         "reduce products issued (x4)", None, "mid tile -> slab + split (once)", "reduce epilogue + store (once)"]
tot = sum(v[i] for i in range(8) if i != 5)
print(f"{n} wave-tails; cycles per wave and tile (100 MHz x ?: s_memtime ticks), share of the tail")
for i, nm in enumerate(names):
    if nm: print(f"  {nm:45s} {v[i] / n:10.0f}  {100.0 * v[i] / tot:5.1f} %")
print(f"  {'sum':45s} {tot / n:10.0f}")
