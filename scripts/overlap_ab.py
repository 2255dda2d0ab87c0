"""A/B of cross-frame overlap (two streams) on the bench workload."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
cols, rows, tris = int(os.environ.get("AR_COLS", 1920)), int(os.environ.get("AR_ROWS", 1080)), int(os.environ.get("AR_TRIS", 1000000))
sc = ar.scenes.random_scene(tris, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for ov in (0, 1, 0, 1):
    R.set_option("overlap", ov)
    for _ in range(20):
        R.render_async(cols, rows)
    R.sync()
    t0 = time.perf_counter(); n = 300
    for _ in range(n):
        R.render_async(cols, rows)
    R.sync()
    dt = time.perf_counter() - t0
    print("overlap=%d  %.1f frames/s  %.1f us/frame" % (ov, n / dt, 1e6 * dt / n), flush=True)
