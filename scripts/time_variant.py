"""Times the frame kernels of a library variant (AR_LIB=path) on the C3 workload; no parity check."""
import importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
tris = int(os.environ.get("AR_TRIS", "1000000")); cols = int(os.environ.get("AR_COLS", "1920")); rows = int(os.environ.get("AR_ROWS", "1080"))
sc = ar.scenes.random_scene(tris, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for _ in range(10): R.render_async(cols, rows)
R.sync()
R.set_profiling(True); R.kernel_stats()
for _ in range(100): R.render_async(cols, rows)
st = R.kernel_stats()
print(os.environ.get("AR_LIB", "default"), {k: round(1e3 * ms / n, 2) for k, (n, ms) in st.items() if n})
