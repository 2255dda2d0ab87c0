"""Frame rate of the blocking ar_render with host output buffers (device->host copies included)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
cols, rows = 1920, 1080
sc = ar.scenes.random_scene(1_000_000, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
rgba = np.zeros((rows, cols, 4), np.uint8); depth = np.zeros((rows, cols), np.float32); asc = np.zeros((rows, cols), np.uint8)
prgba = R.host_array((rows, cols, 4), np.uint8); pdepth = R.host_array((rows, cols), np.float32); pasc = R.host_array((rows, cols), np.uint8)
for label, kw in [("pinned rgba+depth+ascii", dict(rgba=prgba, depth=pdepth, ascii=pasc)), ("pinned rgba+ascii", dict(rgba=prgba, ascii=pasc)),
                  ("pinned ascii only", dict(ascii=pasc)),("rgba+depth+ascii", dict(rgba=rgba, depth=depth, ascii=asc)), ("rgba+ascii", dict(rgba=rgba, ascii=asc)),
                  ("ascii only", dict(ascii=asc))]:
    for _ in range(5):
        R.render_into(cols, rows, **kw)
    t0 = time.perf_counter(); n = 100
    for _ in range(n):
        R.render_into(cols, rows, **kw)
    dt = time.perf_counter() - t0
    print("%-24s %.1f frames/s (%.3f ms/frame)" % (label, n / dt, 1e3 * dt / n), flush=True)
