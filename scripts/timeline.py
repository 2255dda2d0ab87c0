"""Runs C3 frames (three in flight) on the timeline diagnostic build (scripts/make_timeline_variant.py) and prints
when each kernel of each frame started and ended on the GPU's own clock.  AR_LIB = the variant."""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
ar.binding.LIB_PATH = os.environ["AR_LIB"]
tris = int(os.environ.get("AR_TRIS", "1000000")); cols = int(os.environ.get("AR_COLS", "1920")); rows = int(os.environ.get("AR_ROWS", "1080"))
sc = ar.scenes.random_scene(tris, cols, rows, point_lights=bool(os.environ.get("AR_PL")))
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
L = ar.binding.load_library()
for _ in range(30): R.render_async(cols, rows)
R.sync()
assert L.ar_debug_timeline_reset() == 0
n = 90
for _ in range(n): R.render_async(cols, rows)
R.sync()
buf = np.zeros(4096 * 10, dtype=np.uint64)
assert L.ar_debug_timeline_read(buf.ctypes.data_as(ctypes.c_void_p), 4096) == 0
t = buf.reshape(4096, 5, 2)
frames = [f for f in range(4096) if t[f, 3, 1] != 0]
frames.sort(key=lambda f: int(t[f, 0, 0]))
frames = frames[30:]  # steady state
names = ["setup", "binscan", "fill", "tile", "ascii"]
t0 = int(t[frames[0], 0, 0])
us = lambda x: (int(x) - t0) / 100.0
print("frame   " + "   ".join("%-15s" % k for k in names) + "   (start-end, us)")
for f in frames[:14]:
    print("%5d   " % f + "   ".join("%6.1f-%-8.1f" % (us(t[f, k, 0]), us(t[f, k, 1])) for k in range(5)))
dur = {k: np.mean([(int(t[f, i, 1]) - int(t[f, i, 0])) / 100.0 for f in frames]) for i, k in enumerate(names)}
gap = {names[i + 1]: np.mean([(int(t[f, i + 1, 0]) - int(t[f, i, 1])) / 100.0 for f in frames]) for i in range(4)}
lat = np.mean([(int(t[f, 4, 1]) - int(t[f, 0, 0])) / 100.0 for f in frames])
starts = sorted(int(t[f, 0, 0]) for f in frames)
period = (starts[-1] - starts[0]) / 100.0 / (len(starts) - 1)
print("mean duration us", {k: round(v, 1) for k, v in dur.items()})
print("mean gap before  ", {k: round(v, 1) for k, v in gap.items()})
print("frame latency %.1f us, period %.1f us (%.0f frames/s)" % (lat, period, 1e6 / period))
# how much of the time is a k_tile running, and two at once
ev = []
for f in frames:
    ev.append((int(t[f, 3, 0]), 1)); ev.append((int(t[f, 3, 1]), -1))
ev.sort()
cur = 0; last = ev[0][0]; acc = {0: 0, 1: 0, 2: 0, 3: 0}
for x, d in ev:
    acc[min(cur, 3)] += x - last; last = x; cur += d
tot = sum(acc.values())
print("k_tile kernels running at once: " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / tot) for k, v in acc.items()))
