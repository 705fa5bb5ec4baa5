"""Sharding + collectives of the multi-GPU path, world_size 2 on the gloo backend (CPU)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shards_cover_everything():
    from odam_amd import parallel
    for n, ws in [(1000, 8), (7, 3), (5, 8), (0, 2)]:
        spans = [parallel.frame_shard(n, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    counts = [300, 10, 250, 12, 11, 100, 40]
    owners = [parallel.object_shard(counts, r, 3) for r in range(3)]
    assert sorted(sum(owners, [])) == list(range(len(counts)))
    assert owners[0][0] == 0 and owners[1][0] == 2          # the two largest go to different ranks


def test_pack_unpack_roundtrip():
    from odam_amd import parallel
    rs = np.random.RandomState(0)
    per = []
    for n in (0, 3, 35):
        rows = -np.ones((n, 79)); rows[:, :15] = rs.normal(size=(n, 15)).astype(np.float32)
        per.append(rows.tolist())
    blk, cnt = parallel.pack_detections(per)
    assert cnt.tolist() == [0, 3, 30]                        # truncated to 30 like process_frame
    back = parallel.unpack_detections(blk, cnt)
    assert np.array_equal(back[1], np.asarray(per[1]))
    assert np.array_equal(back[2], np.asarray(per[2])[:30])


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from odam_amd import parallel
    n_frames = 11
    full_rows = [[[float(f), float(k)] + [0.5 * f + k] * 13 + [-1.0] * 64 for k in range(f % 4)] for f in range(n_frames)]
    s, e = parallel.frame_shard(n_frames, rank, ws)
    blk, cnt = parallel.pack_detections(full_rows[s:e])
    gb, gc = parallel.allgather_detections(blk, cnt, n_frames, "cpu")
    ok1 = gc.tolist() == [f % 4 for f in range(n_frames)]
    back = parallel.unpack_detections(gb, gc)
    ok2 = all(np.array_equal(back[f], np.asarray(full_rows[f]).reshape(-1, 79)) for f in range(n_frames))
    counts = [30, 5, 22, 9, 14]
    owners = [parallel.object_shard(counts, r, ws) for r in range(ws)]
    local = np.stack([np.full(9, 100.0 + i, np.float32) for i in owners[rank]]) if owners[rank] else np.zeros((0, 9), np.float32)
    allp = parallel.allgather_rows(local, len(counts), owners, "cpu")
    ok3 = np.array_equal(allp[:, 0], 100.0 + np.arange(len(counts)))
    q.put((rank, ok1, ok2, ok3))
    dist.destroy_process_group()


@pytest.mark.parametrize("ws", [2, 3])     # 3: uneven frame shards (4, 4, 3) and object shards (2, 2, 1)
def test_allgather_gloo(ws):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7 * ws) % 2000
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
    assert len(res) == ws and all(ok1 and ok2 and ok3 for _, ok1, ok2, ok3 in res), res


def _worker8(rank, ws, port, q):
    """BASELINE configs[2]'s shape: 1000 frames and 30 objects over 8 ranks (125 frames each; objects 4,4,4,4,4,4,3,3), the
    detections gathered once for the whole scene AND chunk by chunk (256-frame chunks dealt over the ranks, the last one
    ragged: 232 frames = 29 per rank), which is what lets the replicated association of chunk i run under the detection of
    chunk i + 1"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from odam_amd import parallel
    n_frames, n_obj = 1000, 30
    rs = np.random.RandomState(5)
    n_per = rs.randint(0, 36, n_frames)                          # some frames above the 30-row cap, some empty

    def rows_of(f):
        r = -np.ones((n_per[f], 79)); r[:, 0] = f; r[:, 1] = np.arange(n_per[f]) % 8
        r[:, 2:15] = (f * 0.001 + np.arange(n_per[f])[:, None] * 0.01 + np.arange(13)[None] * 0.1).astype(np.float32)
        return r
    want = [rows_of(f)[:30] for f in range(n_frames)]
    s, e = parallel.frame_shard(n_frames, rank, ws)
    gb, gc = parallel.allgather_detections(*parallel.pack_detections([rows_of(f) for f in range(s, e)]), n_frames, "cpu")
    back = parallel.unpack_detections(gb, gc)
    ok1 = (e - s == 125) and len(back) == n_frames and all(np.array_equal(back[f], want[f]) for f in range(n_frames))
    # chunk by chunk
    got, spans = [], parallel.chunk_spans(n_frames, 256)
    for c0, c1 in spans:
        a, b = parallel.frame_shard(c1 - c0, rank, ws)
        cb, cc = parallel.allgather_detections(*parallel.pack_detections([rows_of(f) for f in range(c0 + a, c0 + b)]), c1 - c0, "cpu")
        got += parallel.unpack_detections(cb, cc)
    ok2 = spans == [(0, 256), (256, 512), (512, 768), (768, 1000)] and len(got) == n_frames and \
        all(np.array_equal(got[f], want[f]) for f in range(n_frames))
    counts = list(rs.randint(10, 300, n_obj))
    owners = [parallel.object_shard(counts, r, ws) for r in range(ws)]
    local = np.stack([np.full(33, 7.0 * i, np.float32) for i in owners[rank]])
    allp = parallel.allgather_rows(local, n_obj, owners, "cpu")
    ok3 = sorted(len(o) for o in owners) == [3, 3, 4, 4, 4, 4, 4, 4] and np.array_equal(allp[:, 5], 7.0 * np.arange(n_obj))
    q.put((rank, ok1, ok2, ok3))
    dist.destroy_process_group()


def test_allgather_gloo_world_of_8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 77) % 2000
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(8)]
    for p in procs:
        p.join(timeout=60)
    assert len(res) == 8 and all(ok1 and ok2 and ok3 for _, ok1, ok2, ok3 in res), res


def _chain(z, oracle, chunk, overlap=None, device="cpu", fail_at=None):
    """odam_amd.pipeline.run_scene over the canned-detection sequence of tests/golden/process_tracks.npz with the oracle-backed
    test doubles (tests only): returns everything a rank ends up holding"""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden_process import FakeAssociator
    from test_multi_view_host import OracleFitter
    from odam_amd import pipeline
    from odam_amd.processor import OdamProcess

    class _Det:
        device = "cpu"
    proc = OdamProcess(_Det(), FakeAssociator(), None, None, fitter=OracleFitter(oracle))
    proc.init_sequence(z["K"], 480, 640)
    ids = [int(f) for f in z["img_names"]]
    calls = []

    def detect(f0, f1):
        calls.append((f0, f1))
        return [np.asarray(z[f"det{ids[f]}"], np.float64).reshape(-1, 79) for f in range(f0, f1)]
    stages = {}
    if fail_at is not None:
        inner = proc._process_frame_host      # (what process_frames calls per frame for a process object without the native track store)

        def failing(rgb, frame_id, T_wc, detections=None):
            if frame_id == ids[fail_at]:
                raise RuntimeError("association failed on purpose")
            return inner(rgb, frame_id, T_wc, detections)
        proc._process_frame_host = failing
    out = pipeline.run_scene(proc, len(ids), ids, z["scene_T_wcs"], detect=detect, chunk=chunk, device=device, stages=stages, overlap=overlap)
    assert bool(stages.get("overlapped", False)) == (bool(chunk) and chunk < len(ids) and (overlap is True or (overlap is None and dist.is_initialized() and dist.get_world_size() > 1)))
    first = stages["first_pass"]
    cat = lambda ts: np.concatenate([np.asarray(t).reshape(-1, 82) for t in ts])
    return {"tracks": cat(proc.tracks), "lens": np.array([len(t) for t in proc.tracks]), "calls": np.array(calls),
            "p1": first["params"], "f1": first["fitted"], "qc1": np.asarray(first["bboxes_qc"]), "dl1": np.asarray(first["bboxes_dl"]),
            "merged": cat(out["tracks"]), "mlens": np.array([len(t) for t in out["tracks"]]),
            "p2": out["params"], "f2": out["fitted"], "qc2": np.asarray(out["bboxes_qc"]),
            "pts": np.stack([q.compute_ellipsoid_points()[0] for q in out["quadrics"]])}


def _worker_chain(rank, ws, port, chunk, q, device="cpu"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from conftest import GOLDEN, Oracle
    res = _chain(np.load(os.path.join(GOLDEN, "process_tracks.npz")), Oracle(), chunk, device=device)
    q.put((rank, res))
    dist.destroy_process_group()


def test_run_scene_association_overlapped_with_detection_is_the_same_chain(oracle, golden):
    """pipeline.run_scene(overlap=True): detection + gather of chunk i + 1 on a worker thread under the association of chunk i
    (src/processor.py:306-345 is the loop that runs meanwhile) returns what the serial chain returns, bit for bit, and asks the
    detector for the same spans in the same order."""
    z = golden("process_tracks.npz")
    serial = _chain(z, oracle, 16, overlap=False)
    over = _chain(z, oracle, 16, overlap=True)
    for k in serial:
        assert serial[k].shape == over[k].shape, k
        assert np.array_equal(np.ascontiguousarray(serial[k]).view(np.uint8), np.ascontiguousarray(over[k]).view(np.uint8)), k


def test_run_scene_consumer_failure_stops_the_producer(oracle, golden):
    """An exception in the association loop (src/processor.py:306-345 raising on some frame) must not leave the detection worker
    behind -- blocked on the full queue, or issuing collectives on this rank alone: run_scene re-raises and the thread is gone."""
    import threading
    z = golden("process_tracks.npz")
    with pytest.raises(RuntimeError, match="on purpose"):
        _chain(z, oracle, 4, overlap=True, fail_at=1)      # 4-frame chunks: the producer is several chunks ahead and blocked on put()
    assert not [t for t in threading.enumerate() if t.name == "odam-detect" and t.is_alive()]


def test_bench_gpus_n_refuses_to_run_fewer_ranks():
    """`python bench.py --gpus N` starts N ranks itself (bench.py::launch_ranks); with fewer than N visible devices it must refuse
    (rc 2, nothing on stdout) rather than print a line that says n_gpus = N.  (The launch itself is covered on the GPU:
    tests/test_multirank_gpu.py::test_bench_gpus_2_launches_two_ranks.)"""
    import subprocess
    import sys
    from conftest import REPO
    if torch.cuda.device_count() >= 8:
        pytest.skip("needs a box with fewer than 8 devices")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--frames", "64"], capture_output=True, text=True,
                       timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and r.stdout == "" and "needs 8 visible GPUs" in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])
    # and a WORLD_SIZE that contradicts --gpus is refused too, before anything touches a device
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--frames", "64"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout == ""


@pytest.mark.parametrize("chunk,device", [(0, "cpu"), (16, "cpu"), (16, None)])
def test_run_scene_world_of_2_equals_world_of_1(chunk, device, oracle, golden):
    """The whole driver chain (run_processor.py:70-83) through odam_amd.pipeline.run_scene: two gloo ranks -- each detects its
    share of every chunk, association replayed on both, the RESULTING tracks dealt over the ranks for both fit passes, merge in
    between -- end with what one rank computes alone, bit for bit, on both ranks: track rows and IDs, parameters, fitted flags,
    float64 boxes (they travel as 32-bit word pairs), merged tracks, surface points of the final quadrics (re-sampled from the
    parameters where another rank did the fit)."""
    z = golden("process_tracks.npz")
    one = _chain(z, oracle, chunk)
    assert len(one["lens"]) == int(z["n_tracks"]) and one["f1"].sum() >= 2
    for i in range(int(z["n_tracks"])):      # and the one-rank chain is the reference's track list (float32 detection transport)
        assert np.array_equal(one["tracks"][one["lens"][:i].sum():one["lens"][:i + 1].sum(), :2], z[f"track{i}"][:, :2])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 131 + chunk + (7 if device is None else 0)) % 2000
    # device=None: the collective's staging device is resolved inside run_scene (and, chunked, on its worker thread: ADVICE r5)
    procs = [ctx.Process(target=_worker_chain, args=(r, 2, port, chunk, q, device)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    n = len(z["img_names"])
    spans = [(0, n)] if not chunk else [(c, min(n, c + chunk)) for c in range(0, n, chunk)]
    for r in range(2):
        want_calls = [(c0 + s, c0 + e) for c0, c1 in spans for s, e in [__import__("odam_amd.parallel", fromlist=["x"]).frame_shard(c1 - c0, r, 2)]]
        assert res[r]["calls"].tolist() == [list(c) for c in want_calls]      # a rank detects only its share of every chunk
        for k in one:
            if k == "calls":
                continue
            assert one[k].shape == res[r][k].shape, (r, k)
            assert np.array_equal(np.ascontiguousarray(one[k]).view(np.uint8), np.ascontiguousarray(res[r][k]).view(np.uint8)), (r, k)
