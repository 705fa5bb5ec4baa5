"""The N > 1 code path of the hot path itself on ONE GPU: two ranks share the device (gloo between them) and run the
whole bench step -- detect own frame shard -> all-gather of detections -> fit own objects -> all-gather of fits -- and the
result equals the one-rank run bit for bit.  A third run drives the same collectives through RCCL in a world of one
(backend "nccl"), the call path the 8-GPU driver run takes; a fourth through the library's own RCCL communicator
(include/odam_comm.h), also a world of one."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
ARGS = ["--frames", "96", "--objects", "6", "--steps", "1", "--warmup", "0", "--batch", "16", "--no-cpu-baseline", "--no-config4", "--no-full-chain"]


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
    # ... and through the library's own communicator (include/odam_comm.h: odam_allgather_detections / odam_allgather_f32 on RCCL, the
    # unique id carried by a gloo group), world of one
    d = str(tmp_path / "d.npz")
    _run([sys.executable, bench, "--gpus", "1", "--force-collectives", "--native-collectives", "--backend", "gloo", "--dump", d] + ARGS,
         env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(int(port) + 2), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    za, zb, zc, zd = np.load(a), np.load(b), np.load(c), np.load(d)
    assert za["cnt"].sum() >= 96 * 5                       # the scene weights give real detection load
    for z in (zb, zc, zd):
        assert np.array_equal(za["cnt"], z["cnt"])
        assert np.array_equal(za["blk"].view(np.uint32), z["blk"].view(np.uint32))          # detections: bit for bit
        assert np.array_equal(za["fitted"].view(np.uint32), z["fitted"].view(np.uint32))    # fitted parameters + boxes


def test_whole_chain_two_ranks_equal_one_rank(tmp_path):
    """The chain a user runs (run_processor.py:70-83: detect -> associate -> fit -> merge -> fit) through
    odam_amd.pipeline.run_scene on two ranks sharing the device (gloo) against one rank: 104 frames in batches of 16, so the
    ranks' shards end in a ragged batch of 4 and the one-rank run in one of 8; the association is replayed on both ranks,
    the tracks THAT COME OUT are dealt over the ranks for both fit passes.  With the kernel choice pinned independent of the
    row count (odam_config cg.pin = 1) everything is equal bit for bit: every track row (IDs = positions in the list), both
    passes' parameters and boxes, the merged tracks; both ranks of the pair hold the same result.  A chunked gather (two
    all-gathers per scene) gives the same again."""
    bench = os.path.join(REPO, "bench.py")
    chain = ["--only-chain", "--full-chain", "--frames", "104", "--objects", "4", "--batch", "16", "--streams", "2"]
    env = {"ODAM_CONFIG": "cg.pin=1"}
    a, b, c = (str(tmp_path / n) for n in "abc")
    _run([sys.executable, bench, "--gpus", "1", "--dump-chain", a] + chain, env=env)
    port = str(29300 + os.getpid() % 300)
    two = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    _run(two + ["--master-port", port, bench, "--gpus", "2", "--backend", "gloo", "--one-device", "--dump-chain", b] + chain, env=env)
    _run(two + ["--master-port", str(int(port) + 1), bench, "--gpus", "2", "--backend", "gloo", "--one-device", "--chain-chunk", "64",
                "--dump-chain", c] + chain, env=env)
    za = np.load(a + ".r0.npz")
    assert len(za["track_len"]) >= 10 and za["fitted1"].sum() >= 3 and za["fitted2"].sum() >= 3      # the chain did real work
    assert len(za["merged_len"]) <= len(za["track_len"])
    for path in (b + ".r0.npz", b + ".r1.npz", c + ".r0.npz", c + ".r1.npz"):
        z = np.load(path)
        for k in za.files:
            assert za[k].shape == z[k].shape, (path, k)
            assert np.array_equal(za[k].view(np.uint8), z[k].view(np.uint8)), (path, k)


def test_bench_gpus_2_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE): bench.py::launch_ranks starts the two ranks itself -- fresh child
    processes of a parent that made no HIP call, with the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment the reference's
    init_distributed_mode reads (src/utils/misc.py:390-412) -- and relays rank 0's line: n_gpus = 2, a collective world of 2, and the
    same detections and fits as `--gpus 1`, bit for bit (cg.pin = 1 is the bench's default for every world size)."""
    import json
    bench = os.path.join(REPO, "bench.py")
    a, b = (str(tmp_path / f"{n}.npz") for n in "ab")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ODAM_CONFIG")}
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--dump", a] + ARGS, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, bench, "--gpus", "2", "--one-device", "--backend", "gloo", "--dump", b] + ARGS, cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    l1 = json.loads(one.stdout.strip().splitlines()[-1])
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]      # rank 0's line and nothing else on stdout
    l2 = json.loads(lines[0])
    assert l1["n_gpus"] == 1 and l1["library_config"]["cg.pin"] == 1 and l1["collectives"] is None
    assert l2["n_gpus"] == 2 and l2["library_config"]["cg.pin"] == 1
    assert l2["collectives"]["world"] == 2 and l2["collectives"]["launcher"] == "bench.py" and l2["collectives"]["backend"] == "gloo"
    assert l2["config"]["frames"] == 96 and "cpu_baseline" not in l2
    za, zb = np.load(a), np.load(b)
    assert np.array_equal(za["cnt"], zb["cnt"])
    assert np.array_equal(za["blk"].view(np.uint32), zb["blk"].view(np.uint32))
    assert np.array_equal(za["fitted"].view(np.uint32), zb["fitted"].view(np.uint32))
    # a rank that dies takes the job down with a non-zero code instead of leaving the others in a collective
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--one-device", "--backend", "no-such-backend"] + ARGS, cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
