"""Host-side cost of enqueueing one frame (time for the enqueue loop to return, before any sync)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
tris = int(os.environ.get("AR_TRIS", "1000")); cols, rows = 1920, 1080
sc = ar.scenes.random_scene(tris, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for ov in (1, 0):
    R.set_option("overlap", ov)
    for _ in range(20): R.render_async(cols, rows)
    R.sync()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n): R.render_async(cols, rows)
    t1 = time.perf_counter()
    R.sync()
    t2 = time.perf_counter()
    print("tris %d overlap %d: enqueue %.1f us/frame, total %.1f us/frame" % (tris, ov, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n), flush=True)
