"""Lane load balance of k_tile's per-lane raster on the divergence diagnostic build (scripts/make_divergence_variant.py).
AR_LIB = the variant; AR_SCENE = c3 (default) | c4 | c5; AR_WORK_CLASSES = 0 for lists in entry order."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
ar.binding.LIB_PATH = os.environ["AR_LIB"]
which = os.environ.get("AR_SCENE", "c3")
sc = {"c3": lambda: ar.scenes.random_scene(1000000, 1920, 1080), "c4": lambda: ar.scenes.random_scene(1000000, 3840, 2160),
      "c5": lambda: ar.scenes.random_scene(10000000, 1920, 1080, seed=0xA5C12, point_lights=True)}[which]()
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
if os.environ.get("AR_WORK_CLASSES"):
    R.set_option("work_classes", int(os.environ["AR_WORK_CLASSES"]))
L = ar.binding.load_library()
def read():
    out = np.zeros(6, dtype=np.uint64)
    assert L.ar_debug_divergence(out.ctypes.data_as(ctypes.c_void_p)) == 0
    return out.astype(np.float64)
R.render_async(sc["cols"], sc["rows"]); R.sync()
a = read()
R.render_async(sc["cols"], sc["rows"]); R.sync()
calls, lanes, rows, maxrows, cands, maxcands = read() - a
print("%s work classes %s: %d wavefront calls, %.1f lanes active of 64; rows per lane %.2f, wavefront maximum %.2f (lanes %.0f%% busy in the row pass); "
      "candidates per lane %.2f, wavefront maximum %.2f (%.0f%% busy in the candidate pass)" % (
          which, os.environ.get("AR_WORK_CLASSES", "1"), calls, lanes / calls, rows / lanes, maxrows / calls, 100 * rows / (64 * maxrows),
          cands / lanes, maxcands / calls, 100 * cands / (64 * maxcands)))
