#!/usr/bin/env python3
"""Benchmark of the 3D -> ASCII raster path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one frame of the hot path (clear, vertex transform + raster + z-resolve, shade, ASCII) over
the synthetic workload "1M random triangles, 1920x1080 cells" with mesh, camera and lights already
resident in HBM.  With N > 1 (one process per GPU, launched by torch.distributed.run) the frame is
split into N horizontal strips, each rank renders its strip from its own replica of the mesh and the
ASCII strips are combined by one RCCL all-gather per frame (total work fixed: "strong" scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--tris", type=int, default=N_TRIS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-batch", type=int, default=16, help="frames per RCCL all-gather (multi-GPU runs)")
    ap.add_argument("--force-gather", action="store_true",
                    help="take the strip + all-gather path even with one rank (rehearsal of the multi-GPU loop on one GPU)")
    args = ap.parse_args()

    # stdout carries the one JSON line and nothing else: libraries that write to file descriptor 1 (RCCL prints a
    # version banner there when a communicator is created) are sent to stderr for the rest of the run
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    ar = importlib.import_module("ascii-renderer_amd")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    gather = world > 1 or args.force_gather
    if gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except TypeError:                                  # older torch: no device_id argument
            dist.init_process_group("nccl", rank=rank, world_size=world)

    mg = importlib.import_module("ascii-renderer_amd.multigpu")
    sc = ar.scenes.random_scene(args.tris, args.cols, args.rows)
    R = ar.Renderer(local_rank)
    ar.scenes.upload(R, sc)
    row0, row1 = mg.strip_bounds(args.rows, world, rank)
    if gather:
        R.set_strip(row0, row1)
        torch.cuda.set_stream(torch.cuda.Stream())         # a real stream handle (the default stream's is 0)
        ts = torch.cuda.current_stream().cuda_stream
        # two rings in rotation: the gather of one waits (on its stream) for the frames that filled it while the next
        # frames already fill the other; frames wait for a gather only when they come back to its ring, by which
        # time it is long over: two stream-ordered hand-overs per batch, no host waits, no pipeline drain
        batcher = mg.StripBatcher(row1 - row0, args.cols, max(1, args.gather_batch), "cuda",
                                  before_gather=lambda: R.stream_wait_frame(ts), after_gather=lambda: R.frame_wait_stream(ts))
    R.render_async(args.cols, args.rows)
    R.sync()
    # Multi-GPU loop: frames overlap on the renderer's own streams (three in flight) and write their ASCII
    # strip straight into a slot of the batch ring; every --gather-batch frames ONE RCCL all-gather combines
    # the ranks' rings.  Per frame the host only launches the frame's kernels.

    def step():
        if not gather:
            R.render_async(args.cols, args.rows)
        else:
            R.set_ascii_target(batcher.slot_ptr(), row0)
            R.render_async(args.cols, args.rows)
            batcher.advance()

    def fence():
        if gather:
            batcher.flush()
        R.sync()
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fps = args.steps / dt

    # per-kernel durations (HIP events on the renderer's stream) over a second pass of the same steps
    R.set_profiling(True)
    R.kernel_stats()
    for _ in range(args.steps):
        R.render_async(args.cols, args.rows)
    stats = R.kernel_stats()
    # and once more with the frames strictly one after the other (no overlap between the kernels of
    # consecutive frames): each kernel's duration when it has the GPU to itself
    R.set_option("overlap", 0)
    for _ in range(args.steps):
        R.render_async(args.cols, args.rows)
    stats_serial = R.kernel_stats()
    R.set_option("overlap", 2)
    R.set_profiling(False)
    counters = R.frame_counters()

    gather_ok = None
    if gather:
        # the gathered grid of the last frame must be the full frame: compare with this rank's own full render
        torch.cuda.synchronize()
        R.set_strip(0, 0)
        R.set_ascii_target(None)
        full = torch.from_numpy(R.render(args.cols, args.rows)["ascii"]).cuda()
        gather_ok = bool(torch.equal(full, batcher.frame(batcher.gathered - 1)))
        if not gather_ok:
            print("rank %d: gathered ASCII grid differs from a full-frame render" % rank, file=sys.stderr, flush=True)
    if rank == 0:
        cells = args.cols * (row1 - row0)
        per_launch_bytes = {                             # algorithmic bytes one launch must move
            # tiled path: per-frame triangle stream is the derived position planes (36 B);
            # per cell RGBA8 4 + depth 4 + ramp index 1 written once, visibility keys stay in LDS
            "k_setup": args.tris * 36,
            "k_tile": cells * 9,
            "k_ascii": cells * 2,                        # ramp index in, ASCII code out
            "k_fill": counters["bin_records"] * 4,       # one 4-byte entry per (sub-triangle, bin)
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
        achieved = per_launch_bytes.get(dom, 0) / dom_s / 1e9
        traffic = requests = None
        valu = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic = tj.get(dom)
            requests = tj.get("_read_requests", {}).get(dom)
            valu = tj.get("_valu_instructions", {})
        line = {
            "metric": "frames/sec at 1920x1080 cell grid, 1M tris", "value": fps, "unit": "frames/s",
            "mtri_per_s": fps * args.tris / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32+f64",
            "data": "synthetic",
            "config": {"workload": "random mesh mulberry32(0xA5C11), %d tris, %dx%d cells, camera at origin, default lights, "
                                   "modal smoothing on" % (args.tris, args.cols, args.rows),
                       "partition": "1 GPU" if not gather else "%d row strips, RCCL all-gather of the ASCII strips every %d frames" % (world, max(1, args.gather_batch)),
                       "algorithmic_bytes_per_frame": ar.scenes.algorithmic_bytes(args.tris, args.cols, args.rows),
                       "frame_hbm_frac": ar.scenes.algorithmic_bytes(args.tris, args.cols, args.rows) * fps / 1e9 / HBM_PEAK_GBPS,
                       "queued_records": counters},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # the kernel's reads are scattered lines: memory-side read requests per launch (PMC) over the live
                         # duration, next to the scattered-line rate a microbenchmark reaches on this chip
                         "read_requests": requests,
                         "read_request_rate_G_per_s": (requests / dom_s / 1e9) if requests else None,
                         "scattered_line_ceiling_G_per_s": SCATTERED_LINES_G_PER_S,
                         # what the kernel is bound by in fact: vector-ALU issue.  Wave instructions per launch (PMC) x 4
                         # cycles over the SIMD-cycles of the live duration at the peak clock
                         "valu_instructions": valu.get(dom),
                         "valu_busy_frac": (valu[dom] * 4.0 / (N_SIMDS * PEAK_CLOCK_HZ * dom_s)) if (valu.get(dom) and not gather) else None},
            # the same for the whole frame at the measured frame rate (kernels of three frames share the chip)
            "frame_valu_busy_frac": (sum(valu.get(k, 0) for k in kern_serial) * 4.0 * fps / (N_SIMDS * PEAK_CLOCK_HZ)) if (valu and not gather) else None,  # (counters are of whole frames on one GPU)
            "gather_verified": gather_ok,
            "kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in kern.items()},
            "kernels_avg_us_serial": {k: round(v, 2) for k, v in kern_serial.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ar, sc)
        print(json.dumps(line), file=result_out, flush=True)
    R.close()
    if gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
