"""Phase durations inside k_tile from the stamps variant (AR_LIB=build_variants/stamps.so)."""
import importlib, os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
ar.binding.LIB_PATH = os.environ["AR_LIB"]
which = os.environ.get("AR_SCENE", "c3")     # c3 (default) | c4 | c5 | ico (C2) | cube (C1)
sc = {"c3": lambda: ar.scenes.random_scene(1000000, 1920, 1080), "c4": lambda: ar.scenes.random_scene(1000000, 3840, 2160), "c5": lambda: ar.scenes.random_scene(10000000, 1920, 1080, seed=0xA5C12, point_lights=True), "ico": ar.scenes.icosphere_scene, "cube": ar.scenes.cube_scene}[which]()
R = ar.Renderer(0)
R.set_option("overlap", 0)
ar.scenes.upload(R, sc)
for _ in range(5): R.render_async(sc["cols"], sc["rows"])
R.sync()
L = ar.load_library()
R.set_option("tile_rows", 32)
ntx, nty = (sc["cols"] + 63) // 64, (sc["rows"] + 31) // 32
n_wg = ((((ntx + 1) // 2) * ((nty + 1) // 2) + 7) // 8 * 8) * 4        # k_tile's grid: bins rounded up to 8, four tiles each
n = n_wg * 8
out = np.zeros(n, dtype=np.uint64)
L.ar_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.ar_debug_stamps(out.ctypes.data_as(ctypes.c_void_p), n) == 0
st = out.reshape(n_wg, 8).astype(np.int64)
st = st[st[:, 6] > 0]                       # workgroups beyond the tile grid return before the first stamp
t0 = st[:, 0].min()
names = ["filter", "raster small", "raster big", "compact", "shade", "output"]
for k, nm in enumerate(names):
    d = (st[:, k + 1] - st[:, k]) / 100.0
    print("%-12s mean %6.2f us p50 %6.2f p95 %6.2f max %6.2f" % (nm, d.mean(), np.percentile(d, 50), np.percentile(d, 95), d.max()))
tot = (st[:, 6] - st[:, 0]) / 100.0
print("tile total   mean %6.2f us p95 %6.2f max %6.2f;  start spread %.2f us; kernel span %.2f us" % (
    tot.mean(), np.percentile(tot, 95), tot.max(), (st[:, 0].max() - t0) / 100.0, (st[:, 6].max() - t0) / 100.0))
hw = st[:, 7] & 0xFFFFFFFF
xcc = (st[:, 7] >> 32) & 15
st[:, 7] = st[:, 7] >> 40
print("records per tile mean %.0f max %d" % (st[:, 7].mean(), st[:, 7].max()))
# where the tiles ran: XCD, and the compute unit within it (HW_ID: cu_id bits 11:8, sh_id bit 12, se_id bits 15:13)
cu = (hw >> 8) & 0xFF
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcd %d: %4d tiles, tile total mean %.2f max %.2f, end of last tile %.2f us, distinct CUs %d" % (
            x, m.sum(), tot[m].mean(), tot[m].max(), (st[m, 6].max() - t0) / 100.0, len(set(cu[m].tolist()))))
key = xcc * 256 + cu
per_cu = {}
for k, e, t in zip(key.tolist(), ((st[:, 6] - t0) / 100.0).tolist(), tot.tolist()):
    per_cu.setdefault(k, []).append((e, t))
cnt = np.array([len(v) for v in per_cu.values()])
end = np.array([max(e for e, _ in v) for v in per_cu.values()])
print("CUs used %d; tiles per CU min %d mean %.2f max %d; CU end time mean %.2f p95 %.2f max %.2f" % (
    len(per_cu), cnt.min(), cnt.mean(), cnt.max(), end.mean(), np.percentile(end, 95), end.max()))
for c in sorted(set(cnt.tolist())):
    print("  CUs with %d tiles: %3d, end time mean %.2f max %.2f" % (c, (cnt == c).sum(), end[cnt == c].mean(), end[cnt == c].max()))
rs = (st[:, 2] - st[:, 1]) / 100.0
nrec = st[:, 7]
for lo, hi in ((0, 448), (448, 480), (480, 512), (512, 544), (544, 1000)):
    m = (nrec > lo) & (nrec <= hi)
    if m.any():
        print("records %4d..%4d: %4d tiles, raster small mean %.2f p95 %.2f, tile total mean %.2f" % (lo, hi, m.sum(), rs[m].mean(), np.percentile(rs[m], 95), tot[m].mean()))
