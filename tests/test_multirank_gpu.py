"""The N > 1 code path of the hot path itself on ONE GPU: two ranks share the device (gloo between them) and run the
whole bench step -- detect own frame shard -> all-gather of detections -> fit own objects -> all-gather of fits -- and the
result equals the one-rank run bit for bit.  A third run drives the same collectives through RCCL in a world of one
(backend "nccl"), the call path the 8-GPU driver run takes."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
ARGS = ["--frames", "96", "--objects", "6", "--steps", "1", "--warmup", "0", "--batch", "16", "--no-cpu-baseline", "--no-config4"]


def _run(cmd, env=None):
    r = subprocess.run(cmd, cwd=REPO, env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_two_ranks_on_one_device_equal_one_rank(tmp_path):
    bench = os.path.join(REPO, "bench.py")
    a, b, c = (str(tmp_path / f"{n}.npz") for n in "abc")
    _run([sys.executable, bench, "--gpus", "1", "--dump", a] + ARGS)
    port = str(29600 + os.getpid() % 300)
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", port, bench, "--gpus", "2", "--backend", "gloo", "--one-device", "--dump", b] + ARGS)
    _run([sys.executable, bench, "--gpus", "1", "--force-collectives", "--dump", c] + ARGS,
         env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(int(port) + 1), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    za, zb, zc = np.load(a), np.load(b), np.load(c)
    assert za["cnt"].sum() >= 96 * 5                       # the scene weights give real detection load
    for z in (zb, zc):
        assert np.array_equal(za["cnt"], z["cnt"])
        assert np.array_equal(za["blk"].view(np.uint32), z["blk"].view(np.uint32))          # detections: bit for bit
        assert np.array_equal(za["fitted"].view(np.uint32), z["fitted"].view(np.uint32))    # fitted parameters + boxes
