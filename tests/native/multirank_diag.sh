#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
A="--frames 96 --objects 6 --steps 1 --warmup 0 --batch 16 --no-cpu-baseline --no-config4"
python bench.py --gpus 1 --dump /tmp/a1.npz $A > /dev/null 2>&1
python bench.py --gpus 1 --dump /tmp/a2.npz $A > /dev/null 2>&1
ODAM_CG_BIG_F32=0 python bench.py --gpus 1 --dump /tmp/a0.npz $A > /dev/null 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --one-device --dump /tmp/b.npz $A > /dev/null 2>&1
ODAM_CG_BIG_F32=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --backend gloo --one-device --dump /tmp/b0.npz $A > /dev/null 2>&1
python - <<'PY'
import numpy as np
def cmp(x, y, name):
    a, b = np.load(x), np.load(y)
    d = np.abs(a["blk"] - b["blk"]); nz = np.argwhere(d > 0)
    print(name, "cnt equal", np.array_equal(a["cnt"], b["cnt"]), "blk max diff %.3e" % d.max(), "n differing", len(nz), "frames", sorted(set(nz[:, 0].tolist()))[:20], "cols", sorted(set(nz[:, 2].tolist())) if len(nz) else [],
          "fitted max diff %.3e" % np.abs(a["fitted"] - b["fitted"]).max())
cmp("/tmp/a1.npz", "/tmp/a2.npz", "split mode, 1 rank twice:")
cmp("/tmp/a1.npz", "/tmp/b.npz", "split mode, 1 rank vs 2 ranks:")
cmp("/tmp/a0.npz", "/tmp/b0.npz", "fp32 instr, 1 rank vs 2 ranks:")
PY
