"""What one rank of an N-GPU run does, measured on one GPU: frames/s when only the rows of rank r of N are
rendered (C3 workload, frames overlapped, outputs resident; no gather).  The slowest rank bounds the N-GPU rate."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
mg = importlib.import_module("ascii-renderer_amd.multigpu")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
cols, rows = 1920, 1080
sc = ar.scenes.random_scene(1_000_000, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for world in (1, 2, 4, 8):
    worst = None
    for rank in sorted(set((0, world // 2, world - 1))):
        r0, r1 = mg.strip_bounds(rows, world, rank)
        R.set_strip(r0, r1) if world > 1 else R.set_strip(0, 0)
        for _ in range(20): R.render_async(cols, rows)
        R.sync()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter(); n = 300
            for _ in range(n): R.render_async(cols, rows)
            R.sync()
            best = max(best, n / (time.perf_counter() - t0))
        R.set_profiling(True); R.kernel_stats(); R.set_option("overlap", 0)
        for _ in range(50): R.render_async(cols, rows)
        st = R.kernel_stats(); R.set_profiling(False); R.set_option("overlap", 1)
        us = {k: round(1e3 * ms / n, 1) for k, (n, ms) in st.items() if n}
        print("world %d rank %d rows [%d,%d): %8.0f frames/s   serial kernel us %s" % (world, rank, r0, r1, best, us), flush=True)
        worst = best if worst is None else min(worst, best)
    print("world %d: slowest rank %.0f frames/s (%.2fx one GPU's %s)" % (world, worst, worst / base if world > 1 else 1.0, "rate" ), flush=True) if world > 1 else None
    if world == 1: base = worst
