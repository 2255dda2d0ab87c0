"""Times a library variant (AR_LIB=path) on the C3 workload: frames/s with and without frame overlap,
and per-kernel durations with the frames run one after the other.  No parity check."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
tris = int(os.environ.get("AR_TRIS", "1000000")); cols = int(os.environ.get("AR_COLS", "1920")); rows = int(os.environ.get("AR_ROWS", "1080"))
sc = ar.scenes.random_scene(tris, cols, rows, point_lights=bool(os.environ.get("AR_PL")))
R = ar.Renderer(0)
ar.scenes.upload(R, sc, mode_on=not os.environ.get("AR_NO_SMOOTHING"))
if os.environ.get("AR_TILE_ROWS"):
    R.set_option("tile_rows", int(os.environ["AR_TILE_ROWS"]))
if os.environ.get("AR_FOLD_SCAN"):
    R.set_option("fold_scan", int(os.environ["AR_FOLD_SCAN"]))
if os.environ.get("AR_WORK_CLASSES"):
    R.set_option("work_classes", int(os.environ["AR_WORK_CLASSES"]))
if os.environ.get("AR_STRIP"):      # "world,rank": that rank's strip of the frame's rows
    world, rank = (int(x) for x in os.environ["AR_STRIP"].split(","))
    R.set_strip(*ar.strip_rows(rows, world, rank))
res = {}
chain = int(os.environ.get("AR_CHAIN", "0"))
if chain > 1:
    # frames in chains of `chain` (ar_render_batch_async), a camera per frame as in bench.py: the scene's, turned by a
    # yaw of 0.01 sin(2 pi k / 64) rad at frame k
    import math
    mats = [ar.host.camera_matrices(dict(sc["camera"], yaw=sc["camera"]["yaw"] + 0.01 * math.sin(2.0 * math.pi * k / 64)), cols, rows, 1.0) for k in range(64)]
    packs = [R.pack_cameras([mats[(b * chain + j) % 64] for j in range(chain)]) for b in range(max(1, 64 // chain))]
    cams = packs[0]
    for b in range(3): R.render_batch_async(cols, rows, packs[b % len(packs)])
    R.sync()
    best = 0.0
    for rep in range(5):
        t0 = time.perf_counter(); nb = max(4, 320 // chain)
        for b in range(nb): R.render_batch_async(cols, rows, packs[b % len(packs)])
        R.sync()
        best = max(best, nb * chain / (time.perf_counter() - t0))
    res["fps_chains_of_%d" % chain] = round(best, 1)
    R.set_profiling(True); R.kernel_stats(); R.set_option("overlap", 0)
    for b in range(8): R.render_batch_async(cols, rows, packs[b % len(packs)])
    st = R.kernel_stats(); R.set_profiling(False); R.set_option("overlap", 1)
    res["us_per_frame_in_chains"] = {k: round(1e3 * ms / n / chain, 2) for k, (n, ms) in st.items() if n}
for ov in (1, 0):
    R.set_option("overlap", ov)
    for _ in range(20): R.render_async(cols, rows)
    R.sync()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter(); n = 200
        for _ in range(n): R.render_async(cols, rows)
        R.sync()
        best = max(best, n / (time.perf_counter() - t0))
    res["fps_overlap%d" % ov] = round(best, 1)
R.set_profiling(True); R.kernel_stats()
for _ in range(100): R.render_async(cols, rows)
st = R.kernel_stats()
print(os.path.basename(os.environ.get("AR_LIB", "default")), res, {k: round(1e3 * ms / n, 2) for k, (n, ms) in st.items() if n}, flush=True)
