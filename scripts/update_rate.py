"""In-place vertex updates (ar_update_positions) against a full ar_set_mesh, at the bench mesh: wall time per call and -
under `rocprofv3 --kernel-trace --stats` - the GPU time of the kernels involved.  python scripts/update_rate.py [tris]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ar = importlib.import_module("ascii-renderer_amd")
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cols, rows = 1920, 1080
sc = ar.scenes.random_scene(tris, cols, rows)
R = ar.Renderer(0)
ar.scenes.upload(R, sc)
R.render_async(cols, rows); R.sync()
vbo = np.ascontiguousarray(sc["vbo"], dtype=np.float32).reshape(-1, 9)
pos = R.host_array((vbo.shape[0], 3), np.float32)
pos[...] = vbo[:, 0:3]
whole = R.host_array(vbo.shape, np.float32)
whole[...] = vbo


def timed(label, fn, n=20):
    fn(); R.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        R.render_async(cols, rows)
    R.sync()
    dt = (time.perf_counter() - t0) / n
    print("%-58s %8.3f ms per update + frame" % (label, 1e3 * dt))


t0 = time.perf_counter()
for _ in range(50):
    R.render_async(cols, rows)
R.sync()
print("%-58s %8.3f ms" % ("frame alone (three in flight)", 1e3 * (time.perf_counter() - t0) / 50))
timed("ar_update_positions, positions only (page-locked source)", lambda: R.update_positions(pos))
timed("ar_update_positions, positions + face normals", lambda: R.update_positions(pos, flat_normals=True))
timed("ar_update_positions, whole vertices (page-locked source)", lambda: R.update_positions(whole))
timed("ar_update_positions, positions only (pageable source)", lambda: R.update_positions(vbo[:, 0:3].copy()))
timed("ar_set_mesh (upload, sort, streams)", lambda: R.set_mesh(sc["vbo"], sc["idx"]), n=5)
R.close()
