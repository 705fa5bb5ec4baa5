"""sq_fit_kernel at BASELINE configs[4]'s size (500 objects x 256 views, one workgroup per object): 1024-thread workgroups, one per CU (two rounds of
256) against 512-thread workgroups, two per CU (odam_config sq.two_per_cu), alternating on one box; and whether the two give the same bits.
   python3 tests/native/perf_sq_two_per_cu.py [objects = 500] [views = 256]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from odam_amd import _lib, sq, synth
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 500
views = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"      # view counts spread over 10 ... views (a real scene) instead of all equal
f = sq.SqFitter("cuda:0", 200)
rs = np.random.RandomState(3)
vlist = [int(v) for v in (rs.randint(10, views + 1, n_obj) if mixed else [views] * n_obj)]
cache = {}
def prob(v, s):
    if (v, s) not in cache: cache[(v, s)] = synth.make_sq_problem(v, s)
    return cache[(v, s)]
probs = [prob(vlist[i], i % 16) for i in range(n_obj)]
p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
p0[:, 0] += np.linspace(0, 0.05, n_obj).astype(np.float32)          # not 500 copies of 64 problems
tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
P = np.concatenate([p["P"].astype(np.float32).reshape(-1, 12) for p in probs]); tgt = np.concatenate([t for t, _ in tm]); mask = np.concatenate([m for _, m in tm])
cls = [p["class_id"] for p in probs]
out = {}
ms = {0: [], 1: []}
for rep in range(4):
    for mode in (0, 1):
        _lib.set_config("sq.two_per_cu", mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); r = f.fit(p0, cls, vlist, P, tgt, mask); e1.record()
        torch.cuda.synchronize()
        ms[mode].append(e0.elapsed_time(e1))
        out[mode] = r["params"].cpu().numpy()
_lib.set_config("sq.two_per_cu", 0)
print("%d objects x %s views, 200 steps: sq.two_per_cu = 0 (1024 threads, object b on workgroup b) %s ms | = 1 (%s) %s ms | same bits: %s" % (
    n_obj, ("10 ... %d" % views) if mixed else str(views), " ".join("%.2f" % m for m in ms[0]), "longest first" if mixed else "two 512-thread workgroups per CU", " ".join("%.2f" % m for m in ms[1]), np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32))))
