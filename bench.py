#!/usr/bin/env python3
"""bench.py -- frames/s end to end (DETR detect + super-quadric fit) on a synthetic 640x480 ScanNet-shaped
scene, BASELINE.json configs[1] (1 GPU) / configs[2] (frames sharded over N GPUs + all-gather).

One "step" = one pass of the hot path over one scene:
  detect all frames of this rank's shard (inputs resident in HBM as decoded uint8 [480,640,3] frames; the
  reference transform -- PIL-bilinear resize to 800x1066, /255, normalise -- runs on the device, then the forward in
  batches of 32 on three HIP streams, per-query post-processing on device, threshold + NMS + row packing on the host)
  -> all-gather of detections (N > 1) -> super-quadric fit of this rank's objects, twice (run_processor.py:81-83
  runs optim -> merge -> optim; the merge itself is outside the hot path) -> all-gather of fitted parameters.
value = frames of the WHOLE scene x steps / wall time (max over ranks).  Strong scaling: the scene is fixed.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W          # no WORLD_SIZE in the environment: this process starts the N ranks itself
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W          # ... or an external launcher does (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)

Rank set-up is the reference's env-driven pattern (src/utils/misc.py:390-412: RANK / WORLD_SIZE / LOCAL_RANK -> set_device ->
init_process_group('nccl')); `launch_ranks` below supplies that environment when nobody else has.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from odam_amd import detector as det_mod, multi_view, parallel, sq, synth, transforms, weights  # noqa: E402
from odam_amd.processor import detection_array  # noqa: E402

PEAK_FP32_MFMA = 157.3e12       # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_HBM = 8.0e12                # MI355X_MICROARCH.md, HBM3E
PEAK_BF16_MFMA = 2.5e15          # MI355X_MICROARCH.md, dense v_mfma_f32_32x32x16_bf16
# fp32 layers large enough for the 256-row kernel are multiplied on the bf16 matrix instruction through an exact
# three-way split of both operands, six MFMAs per 16 k: the matrix peak for fp32 work in that mode
PEAK_F32_SPLIT = PEAK_BF16_MFMA / 6
# what the split loop's own instruction mix reaches in a bare kernel on this chip (six v_mfma_f32_16x16x32_bf16 per block, every
# operand re-read from LDS, random data, every CU busy): tests/native/mfma_shape_probe.hip, profiles/r3_mfma_shape_probe.txt
# ("shape 16": 334-337 TFLOP/s of fp32 work at the 2.05 GHz the chip holds under that load).  Quoted NEXT TO `frac`, never instead.
INKERNEL_CEILING_F32_SPLIT = 334e12
# PMC figures cannot be read from inside this process: they come from profiles/pmc_summary.json, which
# tests/native/sum_pmc.py regenerates from the rocprofv3 --pmc passes of tests/native/run_pmc.sh (separate FETCH_SIZE /
# WRITE_SIZE / SQ passes; the gfx950 x2 correction of FETCH_SIZE is applied there).  Missing or stale file -> null.
PMC_SUMMARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_summary.json")


def parity_block():
    """What the committed parity evidence says about the numbers on this line (north star: bit-exact IDs / class labels,
    super-quadric parameters within 1e-4 relative) -- quoted from files the test suite writes and checks, not re-measured here."""
    out = {"ids_and_class_labels": "bit-exact: detections per frame (count, class, NMS order), every association decision, track "
                                   "membership and merge clusters equal the reference run of the whole chain (tests/test_e2e.py on "
                                   "tests/golden/e2e.npz); class arg-max, keep sets and NMS order equal on every detector golden",
           "sq_per_step": "kernel == CPU oracle bit for bit (1 ... 3000 views); oracle vs reference teacher-forced over 200 steps: points "
                          "<= 4e-7, gradient <= 1e-5 of its max, Adam <= 1 ulp (tests/test_sq_oracle.py)"}
    try:
        with open(os.path.join(REPO, "tests", "golden", "sq_survey_table.txt")) as f:
            tail = [l[1:].strip() for l in f.read().splitlines() if l.startswith("#")][1:]
        out["sq_end_state_survey"] = tail
        out["sq_first_divergence"] = [l for l in tail if l.startswith(("first divergence", "with the reference's OWN"))] or None
        out["sq_end_state_note"] = ("free-running 200-step fits of 46 problems against the reference's own fits: NOT all within 1e-4 -- the fit "
                                    "is a chain of arg-extreme decisions and the reference itself moves by more than 1e-4 under 1-2 ulp nudges "
                                    "of its input (s) and across ATen's own CPU dispatch levels; the gate is e <= max(1e-4, 3 s) per problem "
                                    "(tests/sq_survey_lib.py), whole-chain fits are pinned object by object (tests/e2e_lib.py: 25 of 33 "
                                    "within 5e-5, 6 inside the reference's spread, 1 bistable -- classes decided at test time from the stored spread)")
    except OSError:
        out["sq_end_state_survey"] = None
    try:
        # the whole-chain fixture: the reference against ITSELF (its fits re-run under six 1-2 ulp nudges of its input, stored in e2e.npz by
        # make_golden_e2e.py) -- how many of its own fitted objects it reproduces within the north star's 1e-4; beside it what the
        # test suite holds this build to on the same objects (tests/e2e_lib.py decides the class of every object at test time)
        z = np.load(os.path.join(REPO, "tests", "golden", "e2e.npz"))
        rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(np.asarray(b, np.float64)), 1e-3)))
        rows = {}
        for w, name, n in ((1, "track", int(z["n_tracks"])), (2, "merged", int(z["n_merged"]))):
            fitted = [i for i in range(n) if len(z[f"{name}{i}"]) >= 10]
            s_ = [max(rel(p[i], z[f"params{w}"][i]) for p in z[f"spread_params{w}"]) for i in fitted]
            rows[f"pass{w}"] = {"fitted_objects": len(fitted), "reference_within_1e-4_of_itself": int(sum(v <= 1e-4 for v in s_)),
                                "reference_max_self_deviation": max(s_)}
        rows["this_build"] = ("on the reference's own tracks: 16 of 21 (pass 1) and 9 of 12 (pass 2) fits within 5e-5 of the reference's, at most 4 + 2 "
                              "further ones no farther from it than its own nudged runs, at most one per pass bistable (shown by a one-ulp nudge "
                              "that reproduces the reference to 5e-6) -- tests/e2e_lib.py, classes decided at test time, fixture pinned by sha256")
        out["e2e_fixture_reference_vs_itself"] = rows
    except (OSError, KeyError, ValueError):
        out["e2e_fixture_reference_vs_itself"] = None
    return out


def pmc_summary():
    try:
        with open(PMC_SUMMARY) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build_frames(f0, f1, device, seed=0):
    """resident raw frames f0 .. f1-1 of the scene: 32 distinct synthetic 640x480 uint8 RGB frames repeated with a
    per-repeat intensity offset (frame i = base[i % 32] + (i // 32) % 7, the same on every rank), [f1-f0,480,640,3] in
    HBM.  The reference's transform (PIL-bilinear resize to 800x1066, /255, ImageNet normalisation;
    transforms.py:281-290) runs on the device inside the timed step."""
    base = np.stack(list(synth.make_frames(32, seed=seed)))
    out = np.empty((f1 - f0,) + base.shape[1:], np.uint8)
    for k, i in enumerate(range(f0, f1)):
        out[k] = np.clip(base[i % 32].astype(np.int16) + (i // 32) % 7, 0, 255).astype(np.uint8)
    return torch.from_numpy(out).to(device)


def launch_ranks(n, argv, one_device):
    """`python bench.py --gpus N` from a plain shell: start N fresh rank processes of this same command line, one per GPU, with the
    environment an external launcher would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the pattern the
    reference reads in src/utils/misc.py:390-412), wait for all of them, pass rank 0's stdout (the JSON line) through and return
    non-zero if any rank did.  This parent has imported torch and nothing else: it makes no HIP call (device_count() does not
    initialise the GPU), so the children are ordinary child processes of a process that never touched the device -- never an exec
    from one that has.  Fewer than N visible devices is refused unless --one-device lets the ranks share device 0."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not one_device:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have}; refusing to report a run of fewer ranks "
                         f"as n_gpus={n} (use --one-device to let {n} ranks share one GPU for a functional check)\n")
        return 2
    with socket.socket() as s:      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ODAM_BENCH_LAUNCHER="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 16) // n))))
        # ranks other than 0 print nothing on stdout by contract; whatever a library writes there goes to stderr, so stdout is the line alone
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    # a parent that is told to stop (a driver's timeout) takes its ranks with it -- exactly the processes it started, by handle
    import atexit
    import signal

    def stop_children(*_):
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
    atexit.register(stop_children)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, lambda signum, frame: (stop_children(), sys.exit(128 + signum)))
    rc = 0
    alive = list(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.remove(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {r} exited with code {code}; stopping the other ranks\n")
                for o in alive:      # exactly the children this process started, by handle
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--objects", type=int, default=30)
    ap.add_argument("--batch", type=int, default=0, help="frames per forward; 0 = by the number of frames a call covers (Detector.batch_for: 42 up to 300 "
                    "frames, 38 beyond)")
    ap.add_argument("--streams", type=int, default=0, help="concurrent HIP streams; 0 = by shard size")
    ap.add_argument("--chunk", type=int, default=0, help="frames per device->host copy of detections; 0 = 4 batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks (host-resident frames, shard of 8, with association)")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE configs[3] block (R101, bf16, 50 objects)")
    ap.add_argument("--config4-streams", type=int, default=3, help="HIP streams of the configs[3] block")
    ap.add_argument("--config4-batch", type=int, default=76, help="frames per forward of the configs[3] block")
    ap.add_argument("--config4-frames", type=int, default=1000, help="frames of the configs[3] scene")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo + "
                    "--one-device let several ranks share one GPU for a functional check of the N > 1 path")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--force-collectives", action="store_true", help="initialise the process group and run both "
                    "all-gathers even in a world of one (RCCL call path on a single GPU)")
    ap.add_argument("--native-collectives", action="store_true", help="both all-gathers through the library's own RCCL communicator "
                    "(include/odam_comm.h: odam_allgather_detections) instead of torch.distributed")
    ap.add_argument("--dump", default="", help="rank 0 writes the gathered detections and fitted rows of the last step here (.npz)")
    ap.add_argument("--full-chain", action="store_true", default=True, help="secondary block (on by default): the whole driver chain of "
                    "run_processor.py:70-83 on the ranks of this job (detect shard -> all-gather -> association replayed on every rank -> fit of "
                    "the RESULTING tracks -> merge -> fit; odam_amd/pipeline.py)")
    ap.add_argument("--no-full-chain", dest="full_chain", action="store_false")
    ap.add_argument("--chain-frames", type=int, default=0, help="frames of the --full-chain block (0 = --frames)")
    ap.add_argument("--chain-chunk", type=int, default=0, help="frames per all-gather of the --full-chain block (0 = the whole scene)")
    ap.add_argument("--chain-overlap", default="auto", choices=["auto", "on", "off"], help="--full-chain block: association of chunk i under the "
                    "detection + gather of chunk i + 1 (pipeline.run_scene overlap; auto = on for more than one rank and more than one chunk)")
    ap.add_argument("--chain-variants", action="store_true", help="--full-chain block on more than one rank: also time the chunked and the "
                    "chunked + overlapped forms (on one rank they are always timed); off by default there -- the overlapped form issues its collectives "
                    "from a worker thread, which no multi-GPU box has run yet, and a secondary figure must not be able to take the N-rank line down")
    ap.add_argument("--dump-chain", default="", help="every rank r writes tracks, fits and merge result of the --full-chain block to <path>.r<r>.npz")
    ap.add_argument("--only-chain", action="store_true", help="run nothing but the --full-chain block (tests)")
    ap.add_argument("--no-pin", action="store_true", help="leave odam_config cg.pin at 0 (kernel choice by problem size: a rank's result then "
                    "depends on its shard size at the 1e-5 level and an N-rank line is not bit-comparable with the one-rank line)")
    args = ap.parse_args()
    args.batch_given = args.batch > 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.one_device))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: the line would report a rank count nobody asked for")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if ws > 1 or args.force_collectives:
        import torch.distributed as dist
        if ws == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // ws)))   # ranks share the host cores
        import datetime
        # a collective that a peer never reaches must end the job (non-zero exit, the launcher stops the others), not hang it
        tmo = datetime.timedelta(seconds=900)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=torch.device(f"cuda:{local_rank}"), timeout=tmo)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=ws, timeout=tmo)
    device = f"cuda:{local_rank}"
    from odam_amd import _lib
    if not args.no_pin:
        # every size-dependent kernel choice as for a device-filling problem: a frame's result is then the same bits
        # whatever shard or ragged batch it arrives in, so the N-rank lines of a scaling run and the one-rank line compute the same thing
        _lib.set_config("cg.pin", 1)
    native = None
    if args.native_collectives and (ws > 1 or args.force_collectives):
        native = parallel.use_native_collectives(device)
    # who carries the collectives of this line, and over how many ranks (None: a world of one without a process group)
    if ws > 1 or args.force_collectives:
        collectives = {"backend": ("rccl (odam_comm.h communicator)" if native else
                                   {"nccl": "rccl (torch.distributed nccl)"}.get(args.backend, args.backend)),
                       "world": int(native[0]) if native else torch.distributed.get_world_size(),
                       "launcher": os.environ.get("ODAM_BENCH_LAUNCHER", "external (RANK / WORLD_SIZE in the environment)"),
                       "devices": "one shared (--one-device)" if args.one_device else "one per rank"}
    else:
        collectives = None

    coll_dev = device if (args.backend == "nccl" or args.native_collectives) else "cpu"
    N = args.frames
    f0, f1 = parallel.frame_shard(N, rank, ws)
    K = synth.K_SCANNET
    img_size = (synth.IMG_W, synth.IMG_H)

    n_local = f1 - f0
    BATCH = args.batch if args.batch > 0 else "auto"      # what every detect_resident_chunks call of this run is given
    if args.batch <= 0:
        # by the number of frames a call covers (round 6, cg.pin = 1, tests/native/sweep_shard_batch.sh: a 125-frame shard in three equal
        # batches of 42 -- one per stream -- 135 ms against 141 with 32 + 32 + 32 + 29; 1000 frames: 38 per forward, +0.6 % over 32)
        args.batch = det_mod.Detector.batch_for(n_local)
    if args.streams <= 0:
        # the ring kernels hold a CU's whole LDS, so streams mix at workgroup granularity: with three, more CUs are in different
        # phases of different layers at any moment (tests/native/sweep_streams.sh, same box: 2 / 3 / 4 streams = 1006 / 1020 /
        # 1004 frames/s; configs[3]: 1856 / 1932 / 1935)
        args.streams = 3
    args.chunk = args.chunk or 4 * args.batch
    detector = det_mod.Detector(max_batch=max(args.batch, 42) if BATCH == "auto" else args.batch, device=device, n_streams=args.streams)
    detector.bench_batch = BATCH
    # scene weights: calibrated batch-norm statistics + centred heads -> 10-25 detections per frame after NMS, so the
    # host side of the step (threshold, NMS, row packing) runs at the load a trained detector gives it
    detector.load_state_dict(weights.make_state_dict(seed=0, scene=True))
    fitter = sq.SqFitter(device, 200)
    frames = build_frames(f0, f1, device)
    H, W = transforms.target_size(frames.shape[2], frames.shape[1])   # 800 x 1066

    # SQ problems decoupled from the (random-weight) detector output so the fit runs at the stated sizes
    scene = synth.make_scene(N, args.objects, seed=1, min_views=10, max_views=min(300, N))
    tracks = scene["tracks"]
    owners = [parallel.object_shard([len(t) for t in tracks], r, ws) for r in range(ws)]
    mine_all = mine = owners[rank]
    frame_ids = scene["img_names"]

    if args.only_chain:
        blk = full_chain_block(detector, frames, f0, scene, device, coll_dev, args, rank, ws)
        if rank == 0:
            print(json.dumps({"full_chain": blk}))
        if ws > 1 or args.force_collectives:
            torch.distributed.destroy_process_group()
        return

    det_ms = []
    last = {}

    def step(src=None, objs=None, gather=True):
        """one pass of the hot path; src: this rank's frames (default: the resident shard), objs: the objects it fits"""
        src = frames if src is None else src
        mine = mine_all if objs is None else objs
        per_frame = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for c0, rows16 in detector.detect_resident_chunks(src, img_size, K, chunk=args.chunk, batch=BATCH):
            for i in range(rows16.shape[0]):   # host: threshold + NMS + 79-column rows, overlapped with the GPU
                sel = detector.select(rows16[i], 0.6, True, 30)
                objs = {k: [v] for k, v in sel.items()}
                per_frame.append(detection_array(objs, 0, frame_ids[f0 + c0 + i], synth.IMG_W, synth.IMG_H))
        e1.record()          # after the last chunk's copy on the issuing stream: end of the detection phase
        det_ms.append((e0, e1))
        blk, cnt = parallel.pack_detections(per_frame)
        if gather:
            blk, cnt = parallel.allgather_detections(blk, cnt, N, coll_dev, force=args.force_collectives)
        n_det = int(cnt.sum())
        last["blk"], last["cnt"], last["per_frame"] = blk, cnt, per_frame
        fitted = None
        t_fit = time.perf_counter()      # every detection of the step is on the host by now: what follows is the fit phase alone
        for _ in range(2):  # optim -> (merge) -> optim
            out = multi_view.optim_process([tracks[i] for i in mine], frame_ids, scene["T_wcs"], scene["P_cws"],
                                           synth.IMG_H, synth.IMG_W, K, "super_quadric", True, 200, 10,
                                           fitter=fitter, return_params=True)
            local = np.concatenate([out["params"].reshape(len(mine), 9),
                                    np.asarray(out["bboxes_qc"], np.float32).reshape(len(mine), 24)], 1) \
                if mine else np.zeros((0, 33), np.float32)
            fitted = parallel.allgather_rows(local, len(tracks), owners, coll_dev, force=args.force_collectives) if gather else local
        last["fit_ms"] = 1e3 * (time.perf_counter() - t_fit)      # (the passes end with their results on the host)
        return n_det, fitted

    def sync():
        torch.cuda.synchronize()
        if ws > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    del det_ms[:]
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_det, fitted = step()
    sync()
    dt = time.perf_counter() - t0
    dump = dict(last, fitted=fitted)        # the timed region's result (later passes below run other modes)
    det_phase_ms = sum(a.elapsed_time(b) for a, b in det_ms)
    # the contraction kernel alone on the device: one more forward on one stream with an event pair around
    # every launch (flop count per launch comes from the library)
    detector.profile(H, W, True)
    detector(detector.preprocess_u8(frames[:args.batch]))
    torch.cuda.synchronize()
    n_launch, conv_ms, conv_flops = detector.profile_read(H, W)
    att_n, att_ms, att_flops = detector.profile_read_attention(H, W)
    detector.profile(H, W, False)
    # the same scene with every fp32 layer on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32): one untimed + one timed pass
    split_mode = _lib.config()["cg.f32"] == 2
    f32_instr = None
    if split_mode and rank == 0 and ws == 1:
        _lib.check(_lib.lib().odam_op_conv_f32_mode(0), "odam_op_conv_f32_mode")
        step(); torch.cuda.synchronize()
        s0 = time.perf_counter(); step(); torch.cuda.synchronize()
        f32_instr = N / (time.perf_counter() - s0)
        _lib.check(_lib.lib().odam_op_conv_f32_mode(2), "odam_op_conv_f32_mode")
    secondary = {}
    if rank == 0 and ws == 1 and not args.no_secondary:
        secondary = secondary_blocks(step, detector, args.chunk, frames, tracks, N, dump["per_frame"], scene, device, dt / args.steps,
                                     det_phase_ms / args.steps, last)
    if args.dump and rank == 0:
        np.savez(args.dump, blk=dump["blk"], cnt=dump["cnt"], fitted=dump["fitted"])
    if ws > 1:
        t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # stage split (informational): one more untimed scene with explicit timers
    torch.cuda.synchronize(); s0 = time.perf_counter()
    for _ in detector.detect_resident_chunks(frames, img_size, K, chunk=args.chunk, batch=BATCH):
        pass
    torch.cuda.synchronize(); t_det = time.perf_counter() - s0

    result = {
        "metric": "frames/sec end-to-end (DETR detect + SQ fit), 640x480 ScanNet, 1/2/4/8 GPU",
        "not_in_metric": "the association loop between detection and fit (sequential over frames, replicated on every "
                         "rank; DESIGN.md section 5) and the track merge between the two fit passes",
        "value": N * args.steps / dt,
        "unit": "frames/s",
        "n_gpus": ws,
        "rccl_world": collectives["world"] if collectives and collectives["backend"].startswith("rccl") else (1 if ws == 1 else None),
        "collectives": collectives,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "contraction": ("fp32 operands, fp32 accumulation; layers that fill the device multiply through an EXACT three-way bf16 "
                        "split of both operands (a = hi + mid + lo, no rounding) and six products on v_mfma_f32_32x32x16_bf16 "
                        "(dropped terms < 2^-24 |a b|): max error vs float64 0.8-1.4e-7 of sum |a b|, the fp32 matrix instruction's "
                        "own chain 1.1-1.9e-7 (tests/native/x3_probe.hip); every parity test passes unchanged in this mode; "
                        "smaller layers take the same six products on 64- / 128-row tiles (cg.small_x3; layers whose input channels "
                        "are not a multiple of 32 stay on v_mfma_f32_32x32x2_f32)") if split_mode else "v_mfma_f32_32x32x2_f32 (ODAM_CG_BIG_F32=0)",
        "fp32_matrix_instruction_only": ({"value": f32_instr, "unit": "frames/s", "note": "same scene, one timed pass with "
                                          "odam_op_conv_f32_mode(0): every layer on v_mfma_f32_32x32x2_f32"} if f32_instr else None),
        "data": "synthetic (seeded 640x480 uint8 RGB stream resident in HBM, put through the reference transform on the "
                "device inside the step; deterministic random-init DETR-R50 weights; synthetic 82-column object tracks "
                "for the SQ fit)",
        "config": {"workload": f"one scene: {N} frames 640x480 -> 800x1066, DETR-R50 detect + 2 passes of "
                               f"200-step SQ fit over {len(tracks)} objects (10-300 views), fp32",
                   "frames": N, "objects": len(tracks), "detector_batch": args.batch, "detector_batch_rule": ("--batch" if args.batch_given else "Detector.batch_for(frames of a call): 42 up to 300 frames, 38 beyond"),
                   "detector_streams": args.streams,
                   "parallelism": f"frames and objects sharded over {ws} GPU(s)" + (", RCCL all-gather" if ws > 1 else "") +
                                  (" through odam_allgather_detections (include/odam_comm.h)" if args.native_collectives and ws > 1 else "")},
        "library_config": _lib.config(),      # every kernel-selecting switch in force (include/odam_config.h); defaults unless ODAM_CONFIG says otherwise
        "detections_per_scene": n_det,
        "detections_per_frame": n_det / N,
        "detect_only_frames_per_s_per_gpu": (f1 - f0) / t_det,
    }
    result.update(secondary)
    result["parity"] = parity_block()
    pmc = pmc_summary()
    if conv_ms > 0 and det_phase_ms > 0:
        # Batches run on several HIP streams at once, so a launch's own duration includes the kernels it shares
        # the chip with.  `achieved` is therefore the aggregate: algorithmic flops of ALL conv_gemm launches of the
        # timed region / GPU time of the detection phases (HIP events on the issuing stream: first launch -> last
        # copy), which also contains the ~5 % of other kernels -- a lower bound for the kernel.
        flops_per_frame = conv_flops / args.batch
        ach = flops_per_frame * n_local * args.steps / (det_phase_ms * 1e-3)
        iso = conv_flops / (conv_ms * 1e-3)
        peak = PEAK_F32_SPLIT if split_mode else PEAK_FP32_MFMA
        result["roofline"] = {"bound": "mfma", "achieved": ach / 1e12, "peak": peak / 1e12,
                              "unit": "TFLOP/s", "frac": ach / peak,
                              "frac_vs_inkernel_ceiling": (ach / INKERNEL_CEILING_F32_SPLIT) if split_mode else None,
                              "inkernel_ceiling": ({"value": INKERNEL_CEILING_F32_SPLIT / 1e12, "unit": "TFLOP/s",
                                                    "source": "profiles/r3_mfma_shape_probe.txt (tests/native/mfma_shape_probe.hip: the split loop's "
                                                              "instruction mix alone, operands from LDS, random data; the chip holds 2.05 GHz under it)"}
                                                   if split_mode else None),
                              "peak_note": ("dense bf16 MFMA peak / 6 (six bf16 MFMAs per fp32 product block); for scale: the fp32 "
                                            "matrix instruction's peak is 157.3, i.e. achieved / 157.3 = %.2f" % (ach / PEAK_FP32_MFMA))
                              if split_mode else "v_mfma_f32_32x32x2_f32",
                              # HBM-side bytes of this kernel from the rocprofv3 PMC passes recorded in
                              # profiles/pmc_summary.json (per frame; x batch / launches = per launch, like `achieved`)
                              "traffic": (pmc.get("conv_gemm_fp32", {}).get("bytes_per_frame") * args.batch / max(n_launch, 1))
                              if pmc.get("conv_gemm_fp32") else None,
                              "traffic_source": "profiles/pmc_summary.json -- a rocprofv3 --pmc pass the builder ran with tests/native/run_pmc.sh, "
                                                "NOT measured inside this run (counters cannot be read from within the process)",
                              "traffic_unit": "HBM-side bytes per launch, average over the %d launches of a forward of %d "
                                              "frames (PMC: 2 x FETCH_SIZE + WRITE_SIZE; %s)" % (
                                                  n_launch, args.batch, pmc.get("conv_gemm_fp32", {}).get("source", "no PMC summary")),
                              "traffic_bytes_per_frame": pmc.get("conv_gemm_fp32", {}).get("bytes_per_frame"),
                              "algorithmic_gflop_per_launch_avg": conv_flops / max(n_launch, 1) / 1e9,
                              "kernel": "conv_gemm (fp32 implicit-GEMM conv / linear: 256-row LDS-DMA ring kernel with the exact bf16 split "
                                        "for layers that fill the device -- layer1 / layer2 bottlenecks as one launch each: 3x3 + expand + "
                                        "residual, in layer1 also the next reduce --, 128x128 / 64x64 tiles with the same split otherwise)"
                              if split_mode else "conv_gemm_kernel (fp32 implicit-GEMM conv / linear, v_mfma_f32_32x32x2_f32)",
                              "launches_per_forward": n_launch,
                              "algorithmic_gflop_per_frame": flops_per_frame / 1e9,
                              "detection_phase_ms_per_step": det_phase_ms / args.steps,
                              "measured": "aggregate over the timed region: flops of all launches / detection-phase GPU "
                                          "time (HIP events), batches overlapped on %d streams" % args.streams,
                              "single_stream": {"achieved": iso / 1e12, "frac": iso / peak,
                                                "ms_per_forward": conv_ms, "batch": args.batch,
                                                "measured": "one forward alone on one stream, HIP event pair around every launch"}}

    if att_ms > 0:
        att_x3 = _lib.config()["att.x3"] != 0      # fp32 attention through the exact bf16 split (the default)
        att_peak = PEAK_F32_SPLIT if att_x3 else PEAK_FP32_MFMA
        result["attention_kernel"] = {"achieved": att_flops / (att_ms * 1e-3) / 1e12, "peak": att_peak / 1e12,
                                      "unit": "TFLOP/s", "frac": att_flops / (att_ms * 1e-3) / att_peak,
                                      "kernel": ("attention_x3_kernel: fp32 operands, six bf16 matrix instructions per product block"
                                                 if att_x3 else "attention_kernel on v_mfma_f32_32x32x2_f32"),
                                      "launches_per_forward": att_n, "ms_per_forward": att_ms, "batch": args.batch,
                                      "measured": "QK^T + PV flops of the fused attention kernel / its launch durations "
                                                  "(HIP events, one forward alone on one stream)"}
    if mine:
        # the fit kernel alone: three more passes over this rank's objects with an event pair around the launch, the median taken (round 5: a
        # single pass read 8.7 ms on one box and 9.9 on the next; tests/native/perf_sq_fit30.py, seven passes, reads 8.4 on both)
        rec = {"ev": []}
        orig_fit = fitter.fit

        def timed_fit(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig_fit(*a, **k)
            e1.record()
            rec["ev"].append((e0, e1)); rec["views"] = [int(v) for v in a[2]]
            return out

        fitter.fit = timed_fit
        for _ in range(3):
            multi_view.optim_process([tracks[i] for i in mine], frame_ids, scene["T_wcs"], scene["P_cws"], synth.IMG_H,
                                     synth.IMG_W, K, "super_quadric", True, 200, 10, fitter=fitter)
        fitter.fit = orig_fit
        torch.cuda.synchronize()
        if rec["ev"]:
            sq_ms = sorted(a_.elapsed_time(b_) for a_, b_ in rec["ev"])[len(rec["ev"]) // 2]
            views = rec["views"]
            alg = sum(12000 + 80 * v + 40 for v in views) * 200.0   # streaming-formulation bytes per object-step
            sqp = pmc.get("sq_fit", {})
            sq5 = pmc.get("sq_fit_500x256", {})
            result["sq_fit_kernel"] = {
                # The bound that applies: the kernel keeps everything in LDS for all 200 steps, so HBM is not it; its
                # largest phase, the arg-extreme scan of 1000 projected surface points per view, is ~36 vector instructions per point and view of
                # which a third are compares and selects -- so flops against the 157 TFLOP/s vector peak (what only a stream of v_pk_fma_f32
                # reaches) say little, and the honest ceiling is VALU ISSUE: frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x cycles of the
                # launch), the share of the chip's vector issue slots the launch filled (PMC pass on prof_sq.py, see `frac_from_profiles`).
                "bound": "valu-issue", "frac": sqp.get("valu_issue_frac"), "unit": "share of SIMD issue cycles holding a VALU instruction",
                "frac_at_configs4_size": ({"value": sq5.get("valu_issue_frac"), "valu_insts_per_launch": sq5.get("valu_insts_per_launch"),
                                           "note": sq5.get("source")} if sq5 else None),
                "per_wave_view": {"valu_util": sqp.get("valu_util"), "wait_frac": sqp.get("wait_frac"),
                                  "note": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES and SQ_WAIT_ANY / SQ_WAVE_CYCLES: shares of RESIDENT-WAVE time (four waves "
                                          "to a SIMD here), not of the SIMDs' issue slots -- the figure earlier rounds quoted as `frac`"},
                "frac_from_profiles": sqp.get("source"),
                "objects": len(views), "max_views": max(views), "ms_per_launch": sq_ms, "us_per_adam_step": 1e3 * sq_ms / 200,
                "hbm_secondary": {"achieved": alg / (sq_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                                  "frac": alg / (sq_ms * 1e-3) / PEAK_HBM,
                                  "traffic_bytes_per_launch_from_profiles": pmc.get("sq_fit", {}).get("bytes_per_launch"),
                                  "measured": "streaming-formulation bytes (12000 + 80 F + 40 per object-step) x 200 steps / launch "
                                              "duration (HIP events): what an HBM-streaming formulation would move; this kernel "
                                              "sees each view's 80 bytes through L2 only"}}
    if args.full_chain and not args.no_secondary:
        if ws == 1:
            try:
                result["full_chain"] = full_chain_block(detector, frames, f0, scene, device, coll_dev, args, rank, ws)
            except Exception as e:      # noqa: BLE001 -- a secondary figure must not take the line down
                result["full_chain"] = {"error": repr(e)}
        else:       # collective: every rank is inside; an exception on one rank would strand the others, so it is not caught here
            result["full_chain"] = full_chain_block(detector, frames, f0, scene, device, coll_dev, args, rank, ws)
    if rank == 0 and ws == 1 and not args.no_config4:
        del frames
        detector.close()
        torch.cuda.empty_cache()
        result["config4"] = config4_block(device, args, pmc)
    elif rank == 0 and not args.no_config4:
        result["config4"] = {"absent": "BASELINE configs[3] (DETR-R101, 800x600, 50 objects, bf16) is a ONE-GPU configuration: its block is "
                                       "measured by the N = 1 line only (python bench.py --gpus 1)"}
    if ws > 1 or args.force_collectives:
        torch.distributed.destroy_process_group()       # the other ranks are done; the CPU baseline below is rank 0's alone
    if rank == 0 and not args.no_cpu_baseline:
        # every line carries it, the N > 1 lines too: rank 0, after the last collective, on the host cores of this box
        if ws > 1:
            torch.set_num_threads(max(1, min(16, os.cpu_count() or 16)))
        result["cpu_baseline"] = cpu_baseline(scene, tracks, N)
    if rank == 0:
        print(json.dumps(result))


def secondary_blocks(step, detector, chunk, frames, tracks, N, per_frame, scene, device, ms_step_s, det_phase_ms_step, last):
    """Secondary figures of the one-GPU line (none of them is `value`):
    host_resident -- the same step with the frames in pinned HOST memory (PCIe-inclusive: uploads overlapped batch by batch);
    shard_of_8    -- what ONE rank of eight does (125-frame shard + its share of the objects, both fit passes), i.e. the
                     terms that do not shrink with N (fit latency: one workgroup per object, 200 serial steps) as numbers;
    with_association -- the sequential association loop the metric leaves out (processor.py:306-345), timed on a sample of
                     this scene's detections and added to the step (no overlap: an upper bound on its cost)."""
    out = {}
    host = frames.cpu().pin_memory()
    step(src=host); torch.cuda.synchronize()
    s0 = time.perf_counter(); step(src=host); torch.cuda.synchronize()
    out["host_resident"] = {"value": N / (time.perf_counter() - s0), "unit": "frames/s",
                            "note": "frames start in pinned host memory; the uint8 upload (0.92 MB per frame over PCIe) runs on a copy "
                                    "stream under the previous batch's kernels"}
    del host
    f0, f1 = parallel.frame_shard(N, 0, 8)
    objs8 = parallel.object_shard([len(t) for t in tracks], 0, 8)
    sub = frames[f0:f1]
    step(src=sub, objs=objs8, gather=False); torch.cuda.synchronize()
    s0 = time.perf_counter(); step(src=sub, objs=objs8, gather=False); torch.cuda.synchronize()
    t8 = time.perf_counter() - s0
    fit8 = last["fit_ms"]      # the two passes of that step, timed inside it (round 5: was the difference of two separately timed runs, +-1 ms)
    out["shard_of_8"] = {"frames": f1 - f0, "objects": len(objs8), "ms_per_step": 1e3 * t8, "detect_ms": 1e3 * t8 - fit8,
                         "fit_ms_two_passes": fit8,
                         "frames_per_s_if_all_8_ranks_take_this_long": N / t8,
                         "note": "ONE GPU running rank 0's share of an 8-way split (no collective): NOT a scaling measurement -- no run "
                                 "on more than one GPU exists; it prices the per-rank step, including the fit latency that does not "
                                 "shrink when objects are sharded"}
    try:
        out["with_association"] = association_block(per_frame, scene, device, N, ms_step_s)
        fit_ms = 1e3 * ms_step_s - det_phase_ms_step
        out["with_association"]["overlapped"] = association_overlapped(detector, frames, chunk, scene, device, N, fit_ms)
    except Exception as e:      # noqa: BLE001 -- a secondary figure must not take the line down
        out.setdefault("with_association", {})["error"] = repr(e)
    return out


def full_chain_block(detector, frames, f0, scene, device, coll_dev, args, rank, ws):
    """The chain a user of the reference runs (run_processor.py:70-83) on this job's ranks, through odam_amd.pipeline.run_scene:
    every rank detects its frames, the detections are all-gathered, the sequential association loop (processor.py:306-345) is
    replayed on every rank with the hand-written scene association weights (tools/scene_weights.py), the tracks THAT COME OUT are
    dealt over the ranks and fitted, gathered, merged (run_merge.py:79-130) and fitted again.  Timed once after one untimed pass
    over the first frames (handles, first tracks).  Not the headline metric: it contains the replicated association loop."""
    from odam_amd import pipeline
    n = args.chain_frames or args.frames
    ids, T = scene["img_names"], scene["T_wcs"]
    img_size = (synth.IMG_W, synth.IMG_H)
    if args.chain_chunk or n != args.frames:      # a rank's frames are then spread over the scene: keep all of them resident
        frames, f0 = build_frames(0, n, device), 0

    def detect(a, b):
        rows = []
        for c0, rows16 in detector.detect_resident_chunks(frames[a - f0:b - f0], img_size, synth.K_SCANNET, chunk=args.chunk,
                                                          batch=(args.batch if args.batch_given else "auto")):
            for i in range(rows16.shape[0]):
                sel = detector.select(rows16[i], 0.6, True, 30)
                rows.append(detection_array({k: [v] for k, v in sel.items()}, 0, ids[a + c0 + i], synth.IMG_W, synth.IMG_H))
        return rows
    proc, net = _new_process(device, fit_iters=200)
    warm = min(n, 2 * args.batch * max(ws, 1))
    pipeline.run_scene(proc, warm, ids, T, detect=detect, device=coll_dev, force=args.force_collectives)
    proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
    stages = {}
    torch.cuda.synchronize()
    if ws > 1:
        torch.distributed.barrier()
    s0 = time.perf_counter()
    ov = {"auto": None, "on": True, "off": False}[args.chain_overlap]
    out = pipeline.run_scene(proc, n, ids, T, detect=detect, chunk=args.chain_chunk, device=coll_dev, force=args.force_collectives,
                             stages=stages, overlap=ov)
    torch.cuda.synchronize()
    if ws > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - s0
    first = stages.pop("first_pass")
    overlapped = bool(stages.pop("overlapped", False))
    if args.dump_chain:
        cat = lambda ts: np.concatenate([np.asarray(t).reshape(-1, 82) for t in ts]) if len(ts) else np.zeros((0, 82))
        np.savez(f"{args.dump_chain}.r{rank}.npz", tracks=cat(proc.tracks), track_len=[len(t) for t in proc.tracks],
                 params1=first["params"], fitted1=first["fitted"], qc1=np.asarray(first["bboxes_qc"]), dl1=np.asarray(first["bboxes_dl"]),
                 merged=cat(out["tracks"]), merged_len=[len(t) for t in out["tracks"]],
                 params2=out["params"], fitted2=out["fitted"], qc2=np.asarray(out["bboxes_qc"]), dl2=np.asarray(out["bboxes_dl"]))
    res = {"frames": n, "ranks": ws, "value": n / dt, "unit": "frames/s", "ms": 1e3 * dt,
           "stages_ms": {k: 1e3 * v for k, v in stages.items()},
           "tracks": len(proc.tracks), "objects_after_merge": len(out["tracks"]),
           "fitted_pass1": int(np.sum(first["fitted"])), "fitted_pass2": int(np.sum(out["fitted"])),
           "chunk_frames": args.chain_chunk or n, "lost_persistent_launches": net.lost_launches,
           "association_overlapped_with_detection": overlapped,
           "note": "odam_amd.pipeline.run_scene: detect (sharded) -> all-gather -> association replayed on every rank -> fit of the "
                   "resulting tracks (sharded) -> all-gather -> merge -> fit (sharded) -> all-gather; wall clock of rank 0 between barriers"}
    if not args.only_chain and not args.chain_chunk and args.chain_overlap == "auto" and (ws == 1 or args.chain_variants):
        # the same chain in chunks of 4 batches, association of chunk i under detection + gather of chunk i + 1 (pipeline.run_scene
        # overlap=True) and, for comparison, the same chunks one after the other: what chunking + overlap buy on THIS number of ranks
        ck = 4 * args.batch * max(ws, 1)
        for key, flag in (("chunked", False), ("chunked_overlapped", True)):
            proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
            st2 = {}
            torch.cuda.synchronize()
            if ws > 1:
                torch.distributed.barrier()
            s0 = time.perf_counter()
            pipeline.run_scene(proc, n, ids, T, detect=detect, chunk=ck, device=coll_dev, force=args.force_collectives, stages=st2, overlap=flag)
            torch.cuda.synchronize()
            if ws > 1:
                torch.distributed.barrier()
            d2 = time.perf_counter() - s0
            st2.pop("first_pass"); st2.pop("overlapped", None)
            res[key] = {"value": n / d2, "unit": "frames/s", "ms": 1e3 * d2, "chunk_frames": ck, "stages_ms": {k: 1e3 * v for k, v in st2.items()}}
    net.close()
    return res


def _new_process(device, tracks_cap=128, fit_iters=1):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import scene_weights
    from odam_amd import associator
    from odam_amd.processor import OdamProcess
    net = associator.Associator({"GNN_layers": ["self", "cross"] * 4, "self_GNN_layers": ["self", "self"],
                                 "sinkhorn_iterations": 100}, max_tracks=tracks_cap, device=device)
    net.load_state_dict(scene_weights.make_scene_associator_state_dict(2, 8, seed=0))

    class _Det:
        pass
    d = _Det(); d.device = device
    proc = OdamProcess(d, net, None, None, fitter=sq.SqFitter(device, fit_iters))
    proc.init_sequence(synth.K_SCANNET, synth.IMG_H, synth.IMG_W)
    return proc, net


def association_block(per_frame, scene, device, N, ms_step_s, n_sample=200):
    """OdamProcess.process_frame (track preparation -> association network + Sinkhorn on the device -> Hungarian step ->
    attach) over the first n_sample frames' detections of this scene, with the hand-written scene association weights of
    the whole-chain fixture (tools/scene_weights.py).  Sequential over frames and replicated on every rank."""
    proc, net = _new_process(device)
    n = min(n_sample, len(per_frame))
    ids, T = scene["img_names"], scene["T_wcs"]
    proc.process_frames([int(ids[i]) for i in range(min(8, n))], [T[i] for i in range(min(8, n))], per_frame[:min(8, n)])      # warm-up: handles, first tracks
    torch.cuda.synchronize(); s0 = time.perf_counter()
    proc.process_frames([int(ids[i]) for i in range(8, n)], [T[i] for i in range(8, n)], per_frame[8:n])
    torch.cuda.synchronize()
    per = (time.perf_counter() - s0) / max(1, n - 8)
    res = {"ms_per_frame": 1e3 * per, "frames_sampled": n - 8, "live_tracks_at_end": len(proc.tracks),
           "lost_persistent_launches": net.lost_launches,
           "value": N / (ms_step_s + N * per), "unit": "frames/s",
           "note": "detect + fit step of this line + the association loop run after it, nothing overlapped"}
    net.close()
    return res


def association_overlapped(detector, frames, chunk, scene, device, N, fit_ms):
    """detections leave the detector chunk by chunk (on N > 1 ranks: one all-gather per chunk), and the
    sequential association of chunk i runs on a second host thread while the detector works on chunk i + 1.  One GPU here:
    the association's small kernels share the device with the detector's.  value = frames / (overlapped detect + associate
    wall time + the two fit passes of the headline step)."""
    import queue
    import threading
    proc, net = _new_process(device)
    ids, T = scene["img_names"], scene["T_wcs"]
    q = queue.Queue()
    err = []

    def worker():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                c0, rows = item
                proc.process_frames([int(ids[c0 + i]) for i in range(len(rows))], [T[c0 + i] for i in range(len(rows))], rows)
        except Exception as e:      # noqa: BLE001
            err.append(repr(e))
    th = threading.Thread(target=worker)
    torch.cuda.synchronize(); s0 = time.perf_counter()
    th.start()
    for c0, rows16 in detector.detect_resident_chunks(frames, (synth.IMG_W, synth.IMG_H), synth.K_SCANNET, chunk=chunk,
                                                      batch=getattr(detector, "bench_batch", None)):
        rows = []
        for i in range(rows16.shape[0]):
            sel = detector.select(rows16[i], 0.6, True, 30)
            rows.append(detection_array({k: [v] for k, v in sel.items()}, 0, ids[c0 + i], synth.IMG_W, synth.IMG_H))
        q.put((c0, rows))
    t_det = time.perf_counter() - s0
    q.put(None)
    th.join()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - s0
    res = {"detect_and_associate_ms": 1e3 * t_all, "detector_done_after_ms": 1e3 * t_det, "live_tracks_at_end": len(proc.tracks),
           "lost_persistent_launches": net.lost_launches, "value": N / (t_all + fit_ms * 1e-3), "unit": "frames/s",
           "chunk_frames": chunk, "errors": err or None,
           "note": "association of chunk i on a second host thread under the detection of chunk i + 1 (one GPU shared), then the "
                   "two fit passes of the headline step"}
    net.close()
    return res


def config4_block(device, args, pmc):
    """BASELINE configs[3]: DETR-R101, 800x600 input (-> 800x1066), bf16 weights / activations on the bf16 matrix
    instruction, a 50-object scene for the fit.  The same step as the headline line on a scene of its own: 1000 resident raw
    800x600 frames through transform + forward + post-processing + host NMS / rows, then two fit passes over 50 objects whose
    tracks were projected with the 800x600 intrinsics (one untimed pass, one timed; nothing extrapolated)."""
    # 76 frames per forward: the ring kernels run one 256-row tile per CU at a time, and at 76 frames layer3 / layer4 / the
    # encoder are 3.9 / 1.98 rounds of 256 tiles (at 32: 1.64 and 0.84 -- the last round a third empty); one forward alone:
    # 0.529 ms per frame at 76 against 0.598 at 32 (tests/native/perf_batch_sweep.py, profiles/r3_batch_sweep.txt)
    B = args.config4_batch
    n_frames, n_obj = args.config4_frames, 50
    det = det_mod.Detector(backbone="resnet101", max_batch=B, device=device, n_streams=args.config4_streams, dtype="bf16")
    det.load_state_dict(weights.make_state_dict(backbone="resnet101", seed=0, scene=True))
    fitter = sq.SqFitter(device, 200)
    base = np.stack(list(synth.make_frames(32, h=600, w=800, seed=4)))
    frames = torch.from_numpy(np.concatenate([base] * ((n_frames + 31) // 32))[:n_frames]).to(device)
    K = np.array([[722.3, 0.0, 399.5], [0.0, 722.3, 299.5], [0.0, 0.0, 1.0]])      # the ScanNet intrinsics scaled to 800x600
    H, W = transforms.target_size(800, 600)
    scene = synth.make_scene(n_frames, n_obj, seed=2, K=K, img_w=800, img_h=600, min_views=10, max_views=min(300, n_frames))
    ids = scene["img_names"]

    def step():
        n = 0
        for c0, rows16 in det.detect_resident_chunks(frames, (800, 600), K, chunk=4 * B):
            for i in range(rows16.shape[0]):
                sel = det.select(rows16[i], 0.6, True, 30)
                n += len(detection_array({k: [v] for k, v in sel.items()}, 0, c0 + i, 800, 600))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            multi_view.optim_process(scene["tracks"], ids, scene["T_wcs"], scene["P_cws"], 600, 800, K, "super_quadric", True, 200, 10,
                                     fitter=fitter)
        torch.cuda.synchronize()
        return n, t1
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_det, t1 = step()
    t2 = time.perf_counter()
    det.profile(H, W, True)
    det(det.preprocess_u8(frames[:B]))
    torch.cuda.synchronize()
    n_launch, conv_ms, conv_flops = det.profile_read(H, W)
    att_n, att_ms, att_flops = det.profile_read_attention(H, W)
    det.profile(H, W, False)
    det.close(); fitter.close()
    fps_det = n_frames / (t1 - t0)
    att_busy = (pmc.get("mfma_util_bf16_r101") or {}).get("odam_dk::attention_bf16_kernel")
    attention = None
    if att_ms > 0:
        attention = {"achieved": att_flops / (att_ms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                     "frac": att_flops / (att_ms * 1e-3) / PEAK_BF16_MFMA,
                     "matrix_pipe_busy_pmc": att_busy,
                     "kernel": "attention_bf16_kernel (bf16 operands, fp32 softmax; V read transposed from a row-major LDS image, one v_cvt_pk_bf16_f32 "
                               "per pair of probabilities)",
                     "launches_per_forward": att_n, "ms_per_forward": att_ms, "batch": B,
                     "bound_note": "vector issue, not the matrix pipe: per 64-key tile a wave issues 8 matrix instructions (256 cycles of pipe) beside "
                                   "~150 vector instructions of softmax at d_head = 32 (DESIGN 7.3): <= ~0.3 of the pipe at best",
                     "measured": "QK^T + PV flops of the attention launches of one forward / their durations (HIP events, one forward alone on one "
                                 "stream); matrix_pipe_busy_pmc from profiles/pmc_summary.json (SQ_VALU_MFMA_BUSY_CYCLES, not measured in this run)"}
    return {"attention_kernel": attention,
            "workload": "DETR-R101, %d frames 800x600 -> 800x1066, bf16 (fp32 accumulate / softmax / LayerNorm / heads), batches of %d on %d "
                        "streams + 2 passes of 200-step SQ fit over %d objects (fp32)" % (n_frames, B, args.config4_streams, len(scene["tracks"])),
            "dtype": "bf16", "frames_per_s_detect": fps_det, "fit_ms_two_passes": 1e3 * (t2 - t1),
            "frames": n_frames, "extrapolated": False,
            "frames_per_s_scene": n_frames / (t2 - t0),
            "detections_per_frame": n_det / n_frames,
            "roofline": {"bound": "mfma", "achieved": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None,
                         "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                         "frac": conv_flops / (conv_ms * 1e-3) / PEAK_BF16_MFMA if conv_ms > 0 else None,
                         "kernel": "conv_gemm (bf16: 256-row LDS-DMA ring kernel; 3x3 stride 1 through one LDS window per three taps; layer1-3 "
                                   "bottlenecks as one launch each: 3x3 + expand + residual, in layer1 / layer2 also the next reduce; 128x128 "
                                   "tiles for small layers)",
                         "launches_per_forward": n_launch, "ms_per_forward_of_%d" % B: conv_ms,
                         "algorithmic_gflop_per_frame": conv_flops / B / 1e9,
                         "traffic": pmc.get("conv_gemm_bf16_r101", {}).get("bytes_per_frame"),
                         "measured": "one forward of %d frames alone on one stream, HIP event pair around every launch" % B}}


def cpu_baseline(scene, tracks, N):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores: a bounded sample.  The torch
    CPU detector is timed at several thread counts (oversubscription made the all-cores figure the slowest in round 1)
    and the best is reported.  The SQ leg is the reference's own algorithm -- torch autograd + Adam under anomaly detection
    (oracle/sq_autograd_oracle.py) --; the analytic-gradient C oracle (1 thread), an order of magnitude faster, is reported beside it."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import detr_oracle
    from conftest import Oracle
    sd = weights.make_state_dict(seed=0, scene=True)
    t = transforms.get_transforms()
    imgs = [t(f)[0][None] for f in synth.make_frames(6, seed=0)]
    ncpu = os.cpu_count() or 8
    sweep = {}
    keep = torch.get_num_threads()

    def run(xs):
        s_ = time.perf_counter()
        for x in xs:
            out = detr_oracle.detr_forward(sd, x)
            detr_oracle.postprocess(out, (synth.IMG_W, synth.IMG_H), 0.6, synth.K_SCANNET)
        return len(xs) / (time.perf_counter() - s_)
    def rates(xs):
        """per-frame rates: a CPU box shared with the driver jitters by 2x between consecutive frames, so every thread count is
        judged by the MEDIAN of >= 5 single-frame timings and the spread is reported"""
        r = []
        for x in xs:
            r.append(run([x]))
        return r
    per = {}
    for nt in sorted({n for n in (8, 16, 32, min(64, ncpu)) if n <= ncpu}):   # beyond 64 threads the box only gets slower (256: 0.01 frames/s)
        torch.set_num_threads(nt)
        detr_oracle.detr_forward(sd, imgs[0])          # warm-up at this thread count
        per[nt] = rates(imgs[1:6])
        sweep[nt] = float(np.median(per[nt]))
    best = max(sweep, key=sweep.get)
    det_fps = sweep[best]                              # >= 5 frames at the thread count that is reported
    torch.set_num_threads(keep)
    o = Oracle()
    sample = sorted(range(len(tracks)), key=lambda i: len(tracks[i]))[len(tracks) // 2 - 1: len(tracks) // 2 + 2]
    probs = []
    for i in sample:
        pr = synth.make_sq_problem(len(tracks[i]), 500 + i)
        tgt, mask = sq.lines_to_targets(pr["bbox_lines"])
        probs.append((sq.init_params(pr["translate"], pr["angle"], pr["dims"]), pr["P"].astype(np.float32), tgt, mask, pr["class_id"]))
    s = time.perf_counter()
    for p0, P, tgt, mask, cls in probs:
        o.fit(p0, P, tgt, mask, cls)
    sq_ops = len(probs) / (time.perf_counter() - s)
    # the reference's OWN algorithm for the fit: torch CPU autograd + torch.optim.Adam with anomaly detection on at every step
    # (sq_libs.py:432-475, restated in oracle/sq_autograd_oracle.py and pinned bit for bit against the reference's runs); the headline
    # baseline is composed from THIS figure, the analytic-gradient C port is reported beside it
    import sq_autograd_oracle as ag
    torch.set_num_threads(1)      # 9 parameters and a [F, 1000, 3] tensor: more threads only add overhead (and the fixtures were made so)
    t_on = []
    for p0, P, tgt, mask, cls in probs[:2]:
        s = time.perf_counter()
        ag.fit(p0, P, tgt, mask, o.icov[cls].reshape(3, 3), anomaly=True)
        t_on.append(time.perf_counter() - s)
    s = time.perf_counter()
    p0, P, tgt, mask, cls = probs[0]
    ag.fit(p0, P, tgt, mask, o.icov[cls].reshape(3, 3), anomaly=False)
    t_off = time.perf_counter() - s
    torch.set_num_threads(keep)
    sq_ag = len(t_on) / sum(t_on)
    e2e = N / (N / det_fps + 2 * len(tracks) / sq_ag)
    e2e_port = N / (N / det_fps + 2 * len(tracks) / sq_ops)
    return {"value": e2e, "unit": "frames/s", "cores": best, "threads_used": best, "host_logical_cores": ncpu, "kind": "port",      # cores = the threads the reported figure used (detector leg; the SQ leg runs on 1)
            "sample": f"detector oracle (torch CPU fp32, host transform + forward + post-processing): 5 frames at each of "
                      f"{sorted(sweep)} threads, median of the single-frame rates per thread count, the best reported ({best} threads: "
                      f"{det_fps:.3f} frames/s, min {min(per[best]):.3f} max {max(per[best]):.3f}) on a host with "
                      f"{ncpu} logical cores; SQ fit the reference's way (torch autograd + Adam, set_detect_anomaly(True) per step as in "
                      f"sq_libs.py:456; 1 thread): {len(t_on)} median-size objects ({[len(p[1]) for p in probs[:2]]} views) at {sq_ag:.3f} objects/s; "
                      f"composed as N/(N/det + 2*O/sq) for the {N}-frame, {len(tracks)}-object scene",
            "detector_frames_per_s": det_fps, "detector_frames_per_s_by_threads": {str(k): v for k, v in sweep.items()},
            "detector_frames_per_s_spread_by_threads": {str(k): [min(v), max(v)] for k, v in per.items()},
            "sq_objects_per_s_autograd": sq_ag,
            "sq_objects_per_s_autograd_anomaly_off": 1.0 / t_off,
            "sq_objects_per_s": sq_ops,
            "sq_note": "sq_objects_per_s_autograd = the reference's algorithm (oracle/sq_autograd_oracle.py, bit-identical to the reference's own "
                       "fits: tests/test_sq_oracle.py); sq_objects_per_s = the analytic-gradient C port (oracle/sq_oracle.c, 1 thread, %d objects), "
                       "which is NOT how the reference computes and would give value_with_c_port" % len(probs),
            "value_with_c_port": e2e_port}


if __name__ == "__main__":
    main()
