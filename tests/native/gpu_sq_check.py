import sys, ctypes, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from odam_amd import sq, synth
O = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle/libodam_oracle.so"))
fp = ctypes.POINTER(ctypes.c_float)
def P_(x): return x.ctypes.data_as(fp)
icov_all = sq.load_scale_prior()
fitter = sq.SqFitter("cuda:0", 200)
probs = [synth.make_sq_problem(F, seed) for F, seed in [(12,0),(12,1),(16,2),(33,3),(64,4),(100,5),(256,6),(10,7)]]
p0 = np.stack([sq.init_params(p["translate"], p["angle"], p["dims"]) for p in probs])
tm = [sq.lines_to_targets(p["bbox_lines"]) for p in probs]
Pm = np.concatenate([p["P"].astype(np.float32).reshape(-1,12) for p in probs])
tgt = np.concatenate([t for t,_ in tm]); mask = np.concatenate([m for _,m in tm])
vc = [len(p["P"]) for p in probs]; cls=[p["class_id"] for p in probs]
t0=time.time()
out = fitter.fit(p0, cls, vc, Pm, tgt, mask, n_iters=200, want_loss=True, want_traj=True)
torch.cuda.synchronize(); t1=time.time()
print("gpu fit time (incl. first-launch)", t1-t0)
t0=time.time(); out = fitter.fit(p0, cls, vc, Pm, tgt, mask, n_iters=200, want_loss=True, want_traj=True); torch.cuda.synchronize(); print("gpu fit time 2nd", time.time()-t0)
gp = out["params"].cpu().numpy(); gtraj = out["traj"].cpu().numpy(); gl = out["loss"].cpu().numpy(); gpts=out["points"].cpu().numpy()
off=0
for i,p in enumerate(probs):
    F=vc[i]
    po=np.zeros(9,np.float32); traj=np.zeros((200,9),np.float32); ll=np.zeros(200,np.float32)
    O.oracle_sq_fit(P_(p0[i]),F,P_(Pm[off:off+F].copy()),P_(tgt[off:off+F].copy()),P_(mask[off:off+F].copy()),P_(icov_all[cls[i]].copy()),0,200,P_(po),P_(traj),P_(ll))
    pts=np.zeros((1000,3),np.float32); O.oracle_sq_points(P_(po),P_(pts))
    same = np.array_equal(traj.view(np.uint32), gtraj[i].view(np.uint32))
    firstbad = -1
    if not same:
        bad = np.where((traj.view(np.uint32)!=gtraj[i].view(np.uint32)).any(1))[0]; firstbad=bad[0]
    print(f"obj {i} F={F} traj bit-exact={same} first_bad_iter={firstbad} final maxabs={np.abs(po-gp[i]).max():.3e} loss bit-exact={np.array_equal(ll.view(np.uint32), gl[i].view(np.uint32))} pts exact={np.array_equal(pts.view(np.uint32), gpts[i].view(np.uint32))}")
    if not same and firstbad>=0:
        print("   oracle", traj[firstbad]); print("   gpu   ", gtraj[i][firstbad])
    off+=F
