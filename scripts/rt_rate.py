"""Ray backend frame rate: frames enqueued back to back (outputs stay in HBM), and k_raytrace's own duration.
Scenes: the reference's limits filled (32 spheres, 64 triangles, 3 planes, 3 point + 2 directional lights, a third of
the materials mirrors) and the 'unified' scene of tests/golden/ref_raytrace_uniforms.json."""
import importlib, json, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ar = importlib.import_module("ascii-renderer_amd")
from test_gpu_raytrace import _synthetic
from conftest import _rt_uniforms_of
FOVY = 80 * math.pi / 180
R = ar.Renderer(0)
R.set_ascii("@%#*+=-:. ", True, 2, 12)
with open(os.path.join(ROOT, "tests", "golden", "ref_raytrace_uniforms.json")) as fh:
    case = next(c for c in json.load(fh)["cases"] if c["scene"] == "unified" and c["cam_name"] == "orbit")
u_ref, cam_ref = _rt_uniforms_of(case)
cam = {"pos": (7.5, 3.0, 6.5), "yaw": -2.4, "pitch": -0.3, "fovy": FOVY, "pix_aspect": 1.0}
for name, u, c in (("limits filled (32 s, 64 t)", _synthetic(5, 32, 64), cam), ("reference 'unified' scene (10 s, 21 t)", u_ref, cam_ref)):
    R.rt_set_scene(u)
    for cols, rows in ((1920, 1080), (3840, 2160)):
        args = (cols, rows, c["pos"], c["yaw"], c["pitch"], c["fovy"], c["pix_aspect"])
        for _ in range(5): R.rt_render_async(*args)
        R.sync()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): R.rt_render_async(*args)
        R.sync()
        fps = n / (time.perf_counter() - t0)
        R.set_profiling(True); R.kernel_stats()
        for _ in range(50): R.rt_render_async(*args)
        st = R.kernel_stats(); R.set_profiling(False)
        us = {k: 1e3 * ms / cnt for k, (cnt, ms) in st.items() if cnt}
        print("%-40s %4dx%-4d %8.0f frames/s  %7.1f Mrays/s primary   k_raytrace %.1f us, k_ascii %.1f us" %
              (name, cols, rows, fps, fps * cols * rows / 1e6, us.get("k_raytrace", 0), us.get("k_ascii", 0)), flush=True)
