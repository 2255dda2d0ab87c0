"""A frame enqueued on a caller's stream (ar_set_stream) makes no synchronising or allocating call once its buffers
exist, so it can be captured into a HIP graph by the caller and replayed: this script captures one C3 frame (and one
frame of the reference's default scene), replays it, checks the replayed outputs against a normal frame, and times
a replay against the five plain launches.  (A replayed graph repeats the captured camera and parameters; a moving
camera needs a re-capture or a parameter update per frame, which is why the library itself launches kernels.)"""
import ctypes, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ar = importlib.import_module("ascii-renderer_amd")
R = ar.Renderer(0)
# the HIP runtime this process already uses (PyTorch's when it is installed)
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
hip = ctypes.CDLL(path)
vp = ctypes.c_void_p


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_raster_host.json")))
dscene = ref["scenes"]["default_scene"]
dv, di = ar.host.build_geometry(dscene)
dproj, dview = ar.host.camera_matrices({"pos": (0.0, 1.5, 6.0), "yaw": -1.2, "pitch": 0.0}, 96, 36, 0.5)
cases = [("C3: 1M triangles, 1920x1080", ar.scenes.random_scene(1_000_000, 1920, 1080)),
         ("default scene, 96x36", {"vbo": dv, "idx": di, "proj": dproj, "view": dview, "lights": ar.host.pack_lights(dscene), "cols": 96, "rows": 36})]
for name, sc in cases:
    cols, rows = sc["cols"], sc["rows"]
    ar.scenes.upload(R, sc)
    want = R.render(cols, rows)
    stream = vp()
    check(hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1), "hipStreamCreateWithFlags")   # non-blocking
    R.set_stream(stream.value)
    for _ in range(3): R.render_async(cols, rows)          # buffers exist, path chosen
    R.sync()
    graph, gexec = vp(), vp()
    check(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")      # hipStreamCaptureModeGlobal
    R.render_async(cols, rows)
    check(hip.hipStreamEndCapture(stream, ctypes.byref(graph)), "hipStreamEndCapture")
    check(hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0), "hipGraphInstantiate")
    n = 1000
    for _ in range(20): check(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")
    check(hip.hipStreamSynchronize(stream), "sync")
    t0 = time.perf_counter()
    for _ in range(n): hip.hipGraphLaunch(gexec, stream)
    t1 = time.perf_counter()
    check(hip.hipStreamSynchronize(stream), "sync")
    t2 = time.perf_counter()
    got = R.read()
    same = all(np.array_equal(got[k], want[k]) for k in ("rgba", "depth", "ascii"))
    t3 = time.perf_counter()
    for _ in range(n): R.render_async(cols, rows)
    t4 = time.perf_counter()
    R.sync()
    t5 = time.perf_counter()
    print("%-30s graph replay: host %.1f us/frame, %.0f frames/s;  plain launches on the same stream: host %.1f us/frame, %.0f frames/s;  replayed outputs identical: %s" % (
        name, 1e6 * (t1 - t0) / n, n / (t2 - t0), 1e6 * (t4 - t3) / n, n / (t5 - t3), same), flush=True)
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)
    R.set_stream(None)
    hip.hipStreamDestroy(stream)
