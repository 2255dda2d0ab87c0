import importlib, os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
ar.binding.LIB_PATH = os.environ["AR_LIB"]
sc = ar.scenes.random_scene(1000000, 1920, 1080)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
for _ in range(5): R.render_async(1920, 1080)
R.sync()
L = ar.load_library()
n = 1020 * 8
out = np.zeros(n, dtype=np.uint64)
L.ar_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.ar_debug_stamps(out.ctypes.data_as(ctypes.c_void_p), n) == 0
st = out.reshape(1020, 8).astype(np.int64)
for k, nm in enumerate(["load", "edges", "loop", "flush total"]):
    d = st[:, k] / 100.0
    print("%-12s mean %6.2f us p95 %6.2f max %6.2f" % (nm, d.mean(), np.percentile(d, 95), d.max()))
print("records per tile mean %.0f max %d" % (st[:, 4].mean(), st[:, 4].max()))
