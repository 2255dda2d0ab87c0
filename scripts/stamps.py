"""Diagnostic: per-phase wall-clock shares of k_tile from a -DAR_STAMPS build (AR_LIB)."""
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
t0 = st[:, 0].min()
names = ["filter", "raster small", "sweep+reset", "compact", "shade", "output"]
print("kernel span (first stamp0 -> last stamp6): %.1f us" % ((st[:, 6].max() - t0) / 100.0))
print("stamp0 spread: %.1f us" % ((st[:, 0].max() - t0) / 100.0))
for k in range(6):
    d = (st[:, k + 1] - st[:, k]) / 100.0
    print("%-14s mean %6.2f us  p95 %6.2f  max %6.2f" % (names[k], d.mean(), np.percentile(d, 95), d.max()))
tot = (st[:, 6] - st[:, 0]) / 100.0
print("per-tile total mean %.2f p95 %.2f max %.2f" % (tot.mean(), np.percentile(tot, 95), tot.max()))
