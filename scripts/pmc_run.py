"""Renders a few C3 frames (for rocprofv3 --pmc passes; keep it short)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
if os.environ.get("AR_LIB"):
    ar.binding.LIB_PATH = os.environ["AR_LIB"]
tris = int(os.environ.get("AR_TRIS", "1000000")); cols = int(os.environ.get("AR_COLS", "1920")); rows = int(os.environ.get("AR_ROWS", "1080"))
sc = ar.scenes.random_scene(tris, cols, rows)
R = ar.Renderer(0)
if os.environ.get("AR_PATH") == "direct":
    R.set_option("path", 1)
ar.scenes.upload(R, sc)
if os.environ.get("AR_STRIP"):
    world, rank = (int(x) for x in os.environ["AR_STRIP"].split(","))
    R.set_strip(*ar.strip_rows(rows, world, rank))
chain = int(os.environ.get("AR_CHAIN", "0"))
# a camera per frame, as bench.py's timed region has it: the scene's, turned by a yaw of 0.01 sin(2 pi k / 64) rad at frame k
import math
mats = [ar.host.camera_matrices(dict(sc["camera"], yaw=sc["camera"]["yaw"] + 0.01 * math.sin(2.0 * math.pi * k / 64)), cols, rows, 1.0) for k in range(64)]
frame = 0
for _ in range(int(os.environ.get("AR_FRAMES", "5"))):
    if chain > 1:
        R.render_batch_async(cols, rows, R.pack_cameras([mats[(frame + j) % 64] for j in range(chain)]))
        frame += chain
    else:
        R.set_camera(*mats[frame % 64])
        R.render_async(cols, rows)
        frame += 1
R.sync()
print("done", R.frame_counters())
