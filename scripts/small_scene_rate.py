"""Frame rates of the small scenes the reference is normally used with: its default scene at the default 96x36
grid, the cube at 80x40 (C1) and the icosphere of 5120 triangles at 640x360 (C2).  Blocking ar_render with
readback (what a backend's render() does) and frames enqueued back to back (outputs resident)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_raster_host.json")))
scene = ref["scenes"]["default_scene"]
vbo, idx = ar.host.build_geometry(scene)
proj, view = ar.host.camera_matrices({"pos": (0.0, 1.5, 6.0), "yaw": -1.2, "pitch": 0.0}, 96, 36, 0.5)
cases = [("default scene (768 tris) 96x36", {"vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": ar.host.pack_lights(scene), "cols": 96, "rows": 36}),
         ("cube (12 tris) 80x40", ar.scenes.cube_scene()), ("icosphere (5120 tris) 640x360", ar.scenes.icosphere_scene())]
R = ar.Renderer(0)
if os.environ.get("AR_FOLD_SCAN"):
    R.set_option("fold_scan", int(os.environ["AR_FOLD_SCAN"]))
for name, sc in cases:
    ar.scenes.upload(R, sc)
    cols, rows = sc["cols"], sc["rows"]
    for _ in range(50): R.render(cols, rows)
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): R.render(cols, rows)
    blocking = n / (time.perf_counter() - t0)
    # the backend's render(time, framebuffer, state): a blocking frame into the caller's RGBA framebuffer (page-locked, from
    # ar_host_alloc), and the same with the ASCII plane as well
    import numpy as np
    fb, asc = R.host_array((rows, cols, 4), np.uint8), R.host_array((rows, cols), np.uint8)
    drop = {}
    for label, kw in (("rgba", {"rgba": fb}), ("rgba+ascii", {"rgba": fb, "ascii": asc})):
        for _ in range(50): R.render_into(cols, rows, **kw)
        t0 = time.perf_counter()
        for _ in range(n): R.render_into(cols, rows, **kw)
        drop[label] = n / (time.perf_counter() - t0)
    for _ in range(50): R.render_async(cols, rows)
    R.sync()
    t0 = time.perf_counter()
    for _ in range(n): R.render_async(cols, rows)
    R.sync()
    resident = n / (time.perf_counter() - t0)
    R.set_profiling(True); R.kernel_stats(); R.set_option("overlap", 0)
    for _ in range(100): R.render_async(cols, rows)
    st = R.kernel_stats(); R.set_profiling(False); R.set_option("overlap", 1)
    us = {k: round(1e3 * ms / c, 1) for k, (c, ms) in st.items() if c}
    print("%-34s blocking, all planes into new arrays %7.0f frames/s (%.1f us)   into a page-locked framebuffer: rgba %7.0f, rgba + ascii %7.0f   resident %7.0f frames/s   kernels us %s"
          % (name, blocking, 1e6 / blocking, drop["rgba"], drop["rgba+ascii"], resident, us), flush=True)
