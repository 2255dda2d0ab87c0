#!/usr/bin/env python3
"""Benchmark of the 3D -> ASCII raster path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one frame of the hot path (clear, vertex transform + raster + z-resolve, shade, ASCII) over
the synthetic workload "1M random triangles, 1920x1080 cells" with mesh and lights already resident in
HBM and the camera set anew for every frame (a small yaw jitter, as the reference reads its camera live
per frame, js/render/backends/raster.js:284-286).  With N > 1 (one process per GPU) the frame is split
into N horizontal strips, each rank renders its strip from its own replica of the mesh and the ASCII
strips are combined by RCCL all-gathers (total work fixed: "strong" scaling).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the
parent - before it has touched the GPU in any way - runs `python -m torch.distributed.run` as a child
process, relays rank 0's JSON line and exits with the child's status.

Timing: W untimed warm-up frames, then rounds of K frames (when K frames take under 10 ms, a whole number of
K-frame blocks back to back, `frames_per_round` in the line), each round bracketed by a barrier and a full
synchronisation and taken as the MAX over ranks; rounds are repeated until at least 0.25 s has been measured
and the median round is reported (`rounds`), so that a short K makes the result neither a one-millisecond
sample nor a measurement of pipeline fill and drain.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E peak
SCATTERED_LINES_G_PER_S = 36.0  # 128-byte lines per second (1e9) for scattered reads, scripts/micro/fetch_granularity.hip (profiles/r01f_fetch_granularity.txt)
N_SIMDS = 256 * 4            # compute units x SIMDs
PEAK_CLOCK_HZ = 2.4e9        # peak engine clock
N_TRIS = 1_000_000
COLS, ROWS = 1920, 1080
MIN_TIMED_S = 0.25           # rounds of --steps frames are repeated until this much has been measured
MIN_ROUND_S = 0.01           # a round is at least this long: --steps frames, or a whole number of such blocks back to back
MAX_ROUNDS = 400
YAW_JITTER = 0.01            # radians either side of the workload's view direction
N_CAMERAS = 64


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tris", type=int, default=N_TRIS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the latency / readback / per-frame-gather legs")
    ap.add_argument("--gather-batch", type=int, default=16, help="frames per RCCL all-gather (multi-GPU runs)")
    ap.add_argument("--force-gather", action="store_true",
                    help="take the strip + all-gather path even with one rank (rehearsal of the multi-GPU loop on one GPU)")
    ap.add_argument("--frames-per-chain", type=int, default=16,
                    help="frames handed to the library per call (ar_render_batch_async: one chain of kernel launches for all of "
                         "them, a camera each); 1 = one call and one chain per frame (ar_render_async)")
    ap.add_argument("--native-gather", action="store_true",
                    help="one process for all --gpus devices: the library's own group (a host thread and a context per device, "
                         "RCCL all-gathers inside the library) instead of one torch.distributed rank per GPU")
    ap.add_argument("--dry", action="store_true",
                    help="launcher rehearsal without a GPU: the ranks meet over gloo, time nothing and print the line's skeleton")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Starts `--gpus` ranks of this script under torch.distributed.run as a CHILD process and relays rank 0's
    line.  Runs before this process has imported torch or made any HIP call (a process that has initialised the
    GPU must neither exec nor be replaced), and returns the child's exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    out, _ = child.communicate()
    line = None
    for raw in out.decode("utf-8", "replace").splitlines():
        s = raw.strip()
        if s.startswith("{") and s.endswith("}"):
            line = s
        elif s:
            print(raw, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif child.returncode == 0:
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
        return 1
    return child.returncode


def cpu_baseline(ar, sc, budget_s=10.0):
    """The CPU oracle (a scalar port; the reference has no CPU rasterizer) timed on this host."""
    from oracle import oracle as orc
    orc.build()
    n, t0 = 0, time.perf_counter()
    while True:
        rgba, _d, _i = orc.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], sc["cols"], sc["rows"], want_ids=False)
        orc.quantize(rgba, sc["cols"], sc["rows"], ar.scenes.DEFAULT_RAMP, 1, 2, 12)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= 3:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d full frames of the same workload (%d tris, %dx%d), single thread, host has %d logical cpus"
                      % (n, sc["idx"].size // 3, sc["cols"], sc["rows"], os.cpu_count() or 0)}


def dry_run(args, world, rank):
    """The launcher's rehearsal: rendezvous over gloo, one collective, the line's skeleton from rank 0."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seen, top = dist.get_world_size(), float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, top = 1, 1.0
    if rank == 0:
        print(json.dumps({"metric": "frames/sec at 1920x1080 cell grid, 1M tris", "value": None, "unit": "frames/s",
                          "n_gpus": world, "ranks_seen": seen, "max_rank_plus_one": top, "steps": args.steps,
                          "warmup": args.warmup, "dry": True}), flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    native = args.native_gather and not args.dry
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not native:
        return launch_ranks(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != (1 if native else args.gpus):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry:
        return dry_run(args, world, rank)

    # stdout carries the one JSON line and nothing else: libraries that write to file descriptor 1 (RCCL prints a
    # version banner there when a communicator is created) are sent to stderr for the rest of the run
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    ar = importlib.import_module("ascii-renderer_amd")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    gather = world > 1 or args.force_gather or native
    if gather and not native:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except TypeError:                                  # older torch: no device_id argument
            dist.init_process_group("nccl", rank=rank, world_size=world)
    ranks_seen = args.gpus if native else (dist.get_world_size() if gather else 1)
    n_strips = args.gpus if native else world

    mg = importlib.import_module("ascii-renderer_amd.multigpu")
    host = importlib.import_module("ascii-renderer_amd.host")
    sc = ar.scenes.random_scene(args.tris, args.cols, args.rows)
    G = None
    if native:
        # the library's own group: device i renders strip i on a host thread of its own; this process only posts frames
        G = ar.Group(list(range(args.gpus)), max(1, args.gather_batch))
        for member in G.members:
            ar.scenes.upload(member, sc)
        R = G.members[0]
    else:
        R = ar.Renderer(local_rank)
        ar.scenes.upload(R, sc)
    # the frame's camera: the workload's, turned by a yaw within +-YAW_JITTER that changes every frame
    cams = []
    for k in range(N_CAMERAS):
        cam = dict(sc["camera"])
        cam["yaw"] = cam["yaw"] + YAW_JITTER * math.sin(2.0 * math.pi * k / N_CAMERAS)
        cams.append(R.prepare_camera(*host.camera_matrices(cam, args.cols, args.rows, 1.0)))
    # the same cameras as one array (twice over, so that any run of up to N_CAMERAS consecutive frames is a slice of it)
    cam_rows = R.pack_cameras([host.camera_matrices(dict(sc["camera"], yaw=sc["camera"]["yaw"] + YAW_JITTER * math.sin(2.0 * math.pi * k / N_CAMERAS)),
                                                    args.cols, args.rows, 1.0) for k in range(N_CAMERAS)])
    cam_rows = np.ascontiguousarray(np.concatenate([cam_rows, cam_rows], axis=0))
    chain = max(1, min(args.frames_per_chain, N_CAMERAS))
    frame_no = [0]
    row0, row1 = mg.strip_bounds(args.rows, n_strips, rank)
    ts = None
    if gather and not native:
        R.set_strip(row0, row1)
        torch.cuda.set_stream(torch.cuda.Stream())         # a real stream handle (the default stream's is 0)
        ts = torch.cuda.current_stream().cuda_stream

    def make_batcher(batch):
        # two rings in rotation: the gather of one waits (on its stream) for the frames that filled it while the next
        # frames already fill the other; frames wait for a gather only when they come back to its ring, by which
        # time it is long over: two stream-ordered hand-overs per batch, no host waits, no pipeline drain
        return mg.StripBatcher(mg.band_height(args.rows, world), args.cols, max(1, batch), "cuda", rows=args.rows,
                               before_gather=lambda: R.stream_wait_frame(ts), after_gather=lambda: R.frame_wait_stream(ts))

    batcher = make_batcher(args.gather_batch) if (gather and not native) else None
    R.render_async(args.cols, args.rows)
    R.sync()
    # Multi-GPU loop: frames overlap on the renderer's own streams (three in flight) and write their ASCII
    # strip straight into a slot of the batch ring; every --gather-batch frames ONE RCCL all-gather combines
    # the ranks' rings.  Per frame the host sets the camera and launches the frame's kernels.

    def step(b=None):
        if b is G and G is not None:
            G.render_async(args.cols, args.rows, cams[frame_no[0] % N_CAMERAS])
            frame_no[0] += 1
            return
        R.set_camera_prepared(cams[frame_no[0] % N_CAMERAS])
        frame_no[0] += 1
        if b is None:
            R.render_async(args.cols, args.rows)
        else:
            R.set_ascii_target(b.slot_ptr(), row0)
            R.render_async(args.cols, args.rows)
            b.advance()

    def frames(n, b=None, per_chain=1):
        """n frames, a new camera each: one launch chain per frame, or per_chain frames per chain (the group batches by itself)."""
        if per_chain <= 1 or (b is G and G is not None):
            for _ in range(n):
                step(b)
            return
        left = n
        while left > 0:
            m = min(per_chain, left)
            if b is not None:
                m = min(m, b.batch - b.filled)                      # a chain's strips fill consecutive slots of one ring
                R.set_ascii_target(b.slot_ptr(), row0)
            start = frame_no[0] % N_CAMERAS
            R.render_batch_async(args.cols, args.rows, cam_rows[start:start + m], ascii_stride=(b.h * b.cols if b is not None else 0))
            frame_no[0] += m
            left -= m
            if b is not None:
                for _ in range(m):
                    b.advance()

    def fence(b=None):
        if b is G and G is not None:
            G.sync()
            return
        if b is not None:
            b.flush()
        R.sync()
        torch.cuda.synchronize()
        if gather and not native:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_round(n, b=None, per_chain=1):
        """n frames between two fences; seconds, the maximum over the ranks."""
        t0 = time.perf_counter()
        frames(n, b, per_chain)
        fence(b)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def measure(n, b=None, per_chain=1):
        """Rounds until MIN_TIMED_S has been measured; (median seconds per frame, rounds, frames per round).
        A round of n frames that lasts under MIN_ROUND_S is mostly pipeline fill and drain (a frame's latency is
        about two frame times), so short rounds are run as a whole number of n-frame blocks back to back; the
        counts follow from the first block's all-rank maximum, so every rank runs the same."""
        first = timed_round(n, b, per_chain)
        blocks = max(1, int(math.ceil(MIN_ROUND_S / max(first, 1e-9))))
        per_round = n * blocks
        one = first if blocks == 1 else timed_round(per_round, b, per_chain)
        rounds = min(MAX_ROUNDS, max(1, int(math.ceil(MIN_TIMED_S / max(one, 1e-9)))))
        dts = sorted([one] + [timed_round(per_round, b, per_chain) for _ in range(rounds - 1)])
        med = dts[len(dts) // 2] if len(dts) % 2 else 0.5 * (dts[len(dts) // 2 - 1] + dts[len(dts) // 2])
        return med / per_round, len(dts), per_round

    if native:
        batcher = G
    frames(args.warmup, batcher, chain)
    fence(batcher)
    frame_s, rounds, frames_per_round = measure(args.steps, batcher, chain)
    fps = 1.0 / frame_s

    extras = {}
    if chain > 1 and not args.no_extras:
        # the same frames with one call and one chain of launches per frame (what the line reported before frames could share a chain)
        frames(args.warmup, batcher, 1)
        fence(batcher)
        extras["fps_one_chain_per_frame"] = 1.0 / measure(args.steps, batcher, 1)[0]
    gather_ok = None
    if native:
        # the gathered grid of the last frame, as every device holds it, must be the full frame
        last = G.frames_gathered() - 1
        got = [G.read(last, member) for member in range(args.gpus)]
        R.set_strip(0, 0)
        R.set_ascii_target(None)
        R.set_camera_prepared(cams[(frame_no[0] - 1) % N_CAMERAS])
        full = R.render(args.cols, args.rows)["ascii"]
        gather_ok = bool(last >= 0 and all(np.array_equal(g, full) for g in got))
        if not gather_ok:
            print("the group's gathered ASCII grid differs from a full-frame render", file=sys.stderr, flush=True)
        R.set_strip(row0, row1)
    elif gather:
        # the gathered grid of the last frame must be the full frame: compare with this rank's own full render
        torch.cuda.synchronize()
        R.set_strip(0, 0)
        R.set_ascii_target(None)
        R.set_camera_prepared(cams[(frame_no[0] - 1) % N_CAMERAS])
        full = torch.from_numpy(R.render(args.cols, args.rows)["ascii"]).cuda()
        gather_ok = bool(torch.equal(full, batcher.frame(batcher.gathered - 1)))
        if not gather_ok:
            print("rank %d: gathered ASCII grid differs from a full-frame render" % rank, file=sys.stderr, flush=True)
        R.set_strip(row0, row1)
        if not args.no_extras and args.gather_batch != 1:
            # the same loop with one all-gather per frame
            b1 = make_batcher(1)
            frames(args.warmup, b1, 1)
            fence(b1)
            extras["fps_gather_every_frame"] = 1.0 / measure(args.steps, b1, 1)[0]
        R.set_ascii_target(None)

    # per-kernel durations (HIP events on the renderer's streams) over a pass of plain frames
    n_prof = min(args.steps, 300)
    R.set_profiling(True)
    R.kernel_stats()
    for _ in range(n_prof):
        step()
    stats = R.kernel_stats()
    # and once more with the frames strictly one after the other (no overlap between the kernels of
    # consecutive frames): each kernel's duration when it has the GPU to itself
    R.set_option("overlap", 0)
    for _ in range(n_prof):
        step()
    stats_serial = R.kernel_stats()
    # and the chains the timed region runs - `chain` frames per launch - again one chain after the other
    stats_chain = {}
    if chain > 1 and not gather:
        frames(max(chain, n_prof // chain * chain), None, chain)
        stats_chain = R.kernel_stats()
    R.set_profiling(False)
    counters = R.frame_counters()
    if not args.no_extras and rank == 0:
        # latency of a frame: enqueue and wait, one frame at a time (no overlap)
        n_lat = min(args.steps, 200)
        R.sync()
        t0 = time.perf_counter()
        for _ in range(n_lat):
            step()
            R.sync()
        extras["latency_ms_serial"] = 1e3 * (time.perf_counter() - t0) / n_lat
    R.set_option("overlap", 1)
    if not args.no_extras and rank == 0 and not gather:
        # what a caller of the blocking entry point gets: ar_render with the RGBA and ASCII planes copied out
        def readback(rgba, ascii_, n=30):
            R.render_into(args.cols, args.rows, rgba=rgba, ascii=ascii_)
            t0 = time.perf_counter()
            for _ in range(n):
                R.set_camera_prepared(cams[frame_no[0] % N_CAMERAS])
                frame_no[0] += 1
                R.render_into(args.cols, args.rows, rgba=rgba, ascii=ascii_)
            return n / (time.perf_counter() - t0)
        extras["readback_fps"] = {
            "pageable_rgba_ascii": readback(np.zeros((args.rows, args.cols, 4), np.uint8), np.zeros((args.rows, args.cols), np.uint8)),
            "pinned_rgba_ascii": readback(R.host_array((args.rows, args.cols, 4), np.uint8), R.host_array((args.rows, args.cols), np.uint8)),
            "pinned_ascii_only": readback(None, R.host_array((args.rows, args.cols), np.uint8)),
        }

    if not args.no_extras and rank == 0 and not gather:
        # the RGBA -> ASCII entry point (AsciiPass.drawFromBuffer's index half) beside the reference's own CPU twin of it:
        # host grid in, host codes out, blocking - what a caller of ar_quantize sees
        rng = np.random.default_rng(12345)

        def quantize_rate(cols, rows, pinned, n=20):
            if pinned:
                grid = R.host_array((rows, cols, 4), np.uint8)
                grid[...] = rng.integers(0, 256, size=(rows, cols, 4), dtype=np.uint8)
                out = R.host_array((rows, cols), np.uint8)
            else:
                grid = rng.integers(0, 256, size=(rows, cols, 4), dtype=np.uint8)
                out = np.zeros((rows, cols), np.uint8)
            grid[..., 3] = 255
            R.quantize_into(grid, cols, rows, out)
            t0 = time.perf_counter()
            for _ in range(n):
                R.quantize_into(grid, cols, rows, out)
            return n / (time.perf_counter() - t0)
        extras["quantize_fps"] = {"1080p_pageable": quantize_rate(1920, 1080, False), "1080p_pinned": quantize_rate(1920, 1080, True),
                                  "4k_pageable": quantize_rate(3840, 2160, False), "4k_pinned": quantize_rate(3840, 2160, True)}

    if rank == 0:
        if native:
            R.set_ascii_target(None)
        cells = args.cols * (row1 - row0)
        entries = counters["bin_records"]
        per_launch_bytes = {                             # algorithmic bytes one launch must move
            # tiled path: per-frame triangle stream is the derived position planes (36 B);
            # per cell RGBA8 4 + depth 4 + ramp index 1 written once, visibility keys stay in LDS
            "k_setup": args.tris * 36,
            "k_tile": cells * 9,
            "k_ascii": cells * 2,                        # ramp index in, ASCII code out
            "k_fill": entries * 4,                       # one 4-byte entry per (sub-triangle, bin)
            "k_binscan": 0,                                # bookkeeping: no algorithmic traffic of its own
            # direct path: reference layout, 3 x 36 B vertices + 12 B indices per triangle
            "k_raster": args.tris * 120,
            "k_shade": cells * 9,                        # RGBA8 + depth f32 + ASCII per cell
            "k_clear": cells * 8, "k_raster_medium": 0, "k_raster_large": 0,
        }
        # roofline: from the pass in which each kernel has the GPU to itself (with frames overlapped the
        # events of a stream also span the time its kernel waits for workgroup slots held by the others)
        kern = {k: {"launches": n, "avg_us": 1e3 * ms / n} for k, (n, ms) in stats.items() if n}
        kern_serial = {k: 1e3 * ms / n for k, (n, ms) in stats_serial.items() if n}
        dom = max(kern_serial, key=lambda k: kern_serial[k])
        dom_s = kern_serial[dom] * 1e-6
        # the launches of the timed region hold `chain` frames each: the roofline figures are of those launches (their
        # duration with the GPU to itself), the launches that hold one frame are quoted beside them
        launch_frames = chain if (chain > 1 and stats_chain.get(dom, (0, 0))[0]) else 1
        launch_s = (1e-3 * stats_chain[dom][1] / stats_chain[dom][0]) if launch_frames > 1 else dom_s
        achieved = per_launch_bytes.get(dom, 0) * launch_frames / launch_s / 1e9
        # Counter figures (rocprofv3 --pmc passes cannot run inside this process) come from profiles/traffic.json, which
        # carries a fingerprint of the kernel sources it was taken on and the frames per launch of its run: when the
        # sources have changed since, or the launches differ, the line says so and carries none of them.
        traffic = requests = None
        valu, tj, busy = {}, {}, {}
        traffic_stale = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            here = importlib.import_module("ascii-renderer_amd.provenance").csrc_hash()
            traffic_stale = tj.get("_csrc_hash") != here or int(tj.get("_frames_per_launch", 1)) != launch_frames
            if traffic_stale:
                tj = {}
            traffic = tj.get(dom)
            requests = tj.get("_read_requests", {}).get(dom)
            valu = tj.get("_valu_instructions", {})
            busy = tj.get("_busy", {})
        clock_hz = busy.get("clock_ghz_under_load", 0.0) * 1e9 or None     # engine clock under load, from GRBM_GUI_ACTIVE
        kb = busy.get("kernels", {})
        # bytes of the layout the per-frame kernels actually read and write: position planes 36 B/triangle, one 64-byte
        # triangle record and one 4-byte entry (written once, read once) per binned (sub-triangle, bin) pair, RGBA8 +
        # depth + ramp index written and ramp index read + ASCII written per cell
        layout_bytes = args.tris * 36 + entries * (64 + 8) + args.cols * args.rows * (9 + 2)
        measured_frame = sum(v for k, v in tj.items() if not k.startswith("_") and isinstance(v, (int, float)) and k in kern_serial) / launch_frames
        line = {
            "metric": "frames/sec at 1920x1080 cell grid, 1M tris", "value": fps, "unit": "frames/s",
            "mtri_per_s": fps * args.tris / 1e6,
            "n_gpus": args.gpus, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "rounds": rounds,
            "frames_per_round": frames_per_round, "ms_per_step": 1e3 * frame_s,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32+f64",
            "data": "synthetic",
            "config": {"workload": "random mesh mulberry32(0xA5C11), %d tris, %dx%d cells, camera at origin with a per-frame yaw "
                                   "jitter of +-%.2f rad, default lights, modal smoothing on; %s"
                                   % (args.tris, args.cols, args.rows, YAW_JITTER,
                                      ("%d frames (a camera each) per library call and chain of kernel launches" % chain) if chain > 1
                                      else "one library call and one chain of kernel launches per frame"),
                       "frames_per_chain": chain,
                       "partition": "1 GPU" if not gather else "%d row strips, RCCL all-gather of the ASCII strips every %d frames%s"
                                    % (n_strips, max(1, args.gather_batch), " (one process, the library's group)" if native else ""),
                       "algorithmic_bytes_per_frame": ar.scenes.algorithmic_bytes(args.tris, args.cols, args.rows),
                       "reference_layout_hbm_frac": ar.scenes.algorithmic_bytes(args.tris, args.cols, args.rows) * fps / 1e9 / HBM_PEAK_GBPS,
                       "layout_bytes_per_frame": layout_bytes,
                       "layout_hbm_frac": layout_bytes * fps / 1e9 / HBM_PEAK_GBPS,
                       "measured_traffic_bytes_per_frame": measured_frame or None,
                       "traffic_over_layout": (measured_frame / layout_bytes) if (measured_frame and not gather) else None,
                       "queued_records": counters},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # achieved, traffic and the counters below are per launch; a launch holds so many frames
                         "frames_per_launch": launch_frames, "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": per_launch_bytes.get(dom, 0) * launch_frames,
                         # the same kernel launched for one frame (ar_render_async), the GPU to itself
                         "one_frame_per_launch": {"launch_us": dom_s * 1e6, "achieved": per_launch_bytes.get(dom, 0) / dom_s / 1e9,
                                                  "frac": per_launch_bytes.get(dom, 0) / dom_s / 1e9 / HBM_PEAK_GBPS},
                         # the kernel's reads are scattered lines: memory-side read requests per launch (PMC) over the live
                         # duration, next to the scattered-line rate a microbenchmark reaches on this chip
                         "read_requests": requests,
                         "read_request_rate_G_per_s": (requests / launch_s / 1e9) if requests else None,
                         "scattered_line_ceiling_G_per_s": SCATTERED_LINES_G_PER_S,
                         # what the kernel is bound by in fact: vector-ALU issue.  From counters: SQ_ACTIVE_INST_VALU (units of
                         # four cycles of one SIMD) over the SIMD-cycles of the live duration at the clock measured under load
                         # (GRBM_GUI_ACTIVE against kernel durations); and over the cycles its compute units were busy at all
                         "valu_instructions": valu.get(dom),
                         "clock_ghz_under_load": (clock_hz / 1e9) if clock_hz else None,
                         "valu_busy_frac": (kb[dom]["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * clock_hz * launch_s))
                                           if (kb.get(dom) and clock_hz and not gather) else None,
                         "valu_busy_frac_while_cu_busy": (kb[dom]["SQ_ACTIVE_INST_VALU"] * 4.0 / (4.0 * kb[dom]["SQ_BUSY_CU_CYCLES"]))
                                                         if (kb.get(dom) and not gather) else None,
                         # HBM bytes the launch moved in fact (PMC) over its live duration: how near the memory side is
                         "traffic_rate_GB_per_s": (traffic / launch_s / 1e9) if traffic else None},
            "traffic_stale": traffic_stale,
            # the same for the whole frame at the measured frame rate (kernels of three frames share the chip)
            "frame_valu_busy_frac": (sum(kb[k]["SQ_ACTIVE_INST_VALU"] for k in kern_serial if k in kb) / launch_frames * 4.0 * fps / (N_SIMDS * clock_hz))
                                    if (kb and clock_hz and not gather) else None,  # (counters are of whole frames on one GPU)
            # the launches of the timed region hold `frames_per_chain` frames each: their durations with the GPU to itself, per frame
            "kernels_us_per_frame_in_chains": {k: round(1e3 * ms / n / chain, 2) for k, (n, ms) in stats_chain.items() if n} or None,
            "gather_verified": gather_ok,
            "kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in kern.items()},
            "kernels_avg_us_serial": {k: round(v, 2) for k, v in kern_serial.items()},
        }
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ar, sc)
            # the one stage of this path the reference itself runs on the CPU, as measured on the reference's code in the
            # build container (quoted; the reference is never shipped to the GPU box)
            line["cpu_baseline"]["reference_stage"] = {
                "fn": "compositeAndQuantize (js/renderer.js:21-99)", "ms_1080p": 219, "ms_4k": 975,
                "where": "reference code imported under Node 12, 1 core, in the build container; random RGBA grid",
                "gpu_twin": "ar_quantize, see quantize_fps"}
        print(json.dumps(line), file=result_out, flush=True)
    if G is not None:
        G.close()
    else:
        R.close()
    if gather and not native:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
