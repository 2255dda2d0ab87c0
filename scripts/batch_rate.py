"""Frames per second when several frames share one chain of launches (ar_render_batch_async) against one chain per frame
(ar_render_async), outputs resident, a camera per frame: the small scenes the reference is used with, and the strip one of
eight GPUs renders of the bench frame (C3).  python scripts/batch_rate.py"""
import importlib, json, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ar = importlib.import_module("ascii-renderer_amd")
mg = importlib.import_module("ascii-renderer_amd.multigpu")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_raster_host.json")))
scene = ref["scenes"]["default_scene"]
vbo, idx = ar.host.build_geometry(scene)
cam0 = {"pos": (0.0, 1.5, 6.0), "yaw": -1.2, "pitch": 0.0}
proj, view = ar.host.camera_matrices(cam0, 96, 36, 0.5)
cases = [("default scene (768 tris) 96x36", {"vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": ar.host.pack_lights(scene), "cols": 96, "rows": 36, "camera": cam0}, 0.5, None),
         ("cube (12 tris) 80x40, C1", ar.scenes.cube_scene(), 0.5, None),
         ("icosphere (5120 tris) 640x360, C2", ar.scenes.icosphere_scene(), 0.5, None)]
c3 = ar.scenes.random_scene(1_000_000, 1920, 1080)
cases.append(("C3 whole frame (1M tris, 1920x1080)", c3, 1.0, None))
for world in (2, 4, 8):
    cases.append(("C3, strip %d of %d" % (world // 2, world), c3, 1.0, mg.strip_bounds(1080, world, world // 2)))
R = ar.Renderer(0)
last = None
for name, sc, aspect, strip in cases:
    if sc is not last:
        ar.scenes.upload(R, sc)
        last = sc
    cols, rows = sc["cols"], sc["rows"]
    R.set_strip(*(strip or (0, 0)))
    cams = []
    for k in range(64):
        cam = dict(sc["camera"])
        cam["yaw"] = cam["yaw"] + 0.01 * math.sin(2.0 * math.pi * k / 64)
        cams.append(ar.host.camera_matrices(cam, cols, rows, aspect))
    prepared = [R.prepare_camera(p, v) for p, v in cams]
    total = 1920 if cols < 1000 else 384

    def single():
        for k in range(total):
            R.set_camera_prepared(prepared[k % 64])
            R.render_async(cols, rows)
        R.sync()

    def batched(n):
        packs = [R.pack_cameras([cams[(b * n + j) % 64] for j in range(n)]) for b in range(max(1, 64 // n))]
        def run():
            for b in range(total // n):
                R.render_batch_async(cols, rows, packs[b % len(packs)])
            R.sync()
        return run

    out = []
    for label, fn in [("1", single)] + [(str(n), batched(n)) for n in (2, 4, 8, 16, 32)]:
        fn()
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            best = max(best, total / (time.perf_counter() - t0))
        out.append("%s: %6.0f" % (label, best))
    print("%-38s frames/s by frames per launch chain   %s" % (name, "   ".join(out)), flush=True)
R.close()
