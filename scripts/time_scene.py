"""Frame rate and kernel times for a coherent mesh (icosphere, subdivision AR_SUBDIV) at AR_COLS x AR_ROWS."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
sub = int(os.environ.get("AR_SUBDIV", "8")); cols = int(os.environ.get("AR_COLS", "1920")); rows = int(os.environ.get("AR_ROWS", "1080"))
sc = ar.scenes.icosphere_scene(angle=0.6, cols=cols, rows=rows, pixel_aspect=1.0, subdiv=sub)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for ov in (1, 0):
    R.set_option("overlap", ov)
    for _ in range(20): R.render_async(cols, rows)
    R.sync()
    t0 = time.perf_counter(); n = 300
    for _ in range(n): R.render_async(cols, rows)
    R.sync()
    print("icosphere %d (%d triangles) %dx%d overlap %d: %.0f frames/s" % (sub, sc["idx"].size // 3, cols, rows, ov, n / (time.perf_counter() - t0)), flush=True)
R.set_profiling(True); R.kernel_stats()
for _ in range(50): R.render_async(cols, rows)
print({k: round(1e3 * ms / n, 2) for k, (n, ms) in R.kernel_stats().items() if n}, R.frame_counters())
