"""Randomised parity sweep: many small scenes with random geometry (soups of tiny to screen-filling
triangles, duplicated and coplanar triangles, shared vertices with smooth normals, NaN / out-of-range
input), cameras (also inside the geometry), grids, lights, ramps, smoothing parameters, strips,
both resolve paths, random atlas texturing (texel coordinates, sample bits, atlases of every alpha class and
of degenerate sizes), meshes built on the GPU from indexed input, and ray-backend scenes, each compared bit
for bit with the oracle.   usage: fuzz_parity.py [cases] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ar = importlib.import_module("ascii-renderer_amd")
from oracle import oracle as orc


def make_case(rng):
    kind = rng.integers(0, 5)
    if kind <= 2:      # soup
        T = int(rng.choice([1, 7, 300, 3000, 20000]))
        centre = rng.uniform(-8, 8, size=(T, 1, 3))
        scale = rng.choice([0.02, 0.2, 2.0, 25.0], size=(T, 1, 1), p=[0.4, 0.3, 0.2, 0.1])
        P = (centre + rng.uniform(-1, 1, size=(T, 3, 3)) * scale).reshape(-1, 3).astype(np.float32)
        I = np.arange(T * 3, dtype=np.uint32).reshape(-1, 3)
        if T > 10 and rng.random() < 0.5:   # exact duplicates and reversed duplicates: depth ties, two-sided surfaces
            k = T // 3
            P[3 * (T - k):] = P[:3 * k]
            if rng.random() < 0.5:
                I[T - k:] = I[T - k:, ::-1]
        C = rng.uniform(0, 1, size=(T, 3)).astype(np.float32)
        vbo, idx = ar.host.mesh_to_vbo(P, I, C)
    elif kind == 3:    # smooth indexed mesh
        P, I = ar.geometry.icosphere(int(rng.integers(0, 5)), float(rng.uniform(0.3, 3.0)))
        vbo, idx = ar.host.mesh_to_vbo(P, I)
        vbo = vbo.reshape(-1, 9).copy()
        # weld to per-vertex normals (smooth shading path: normals differ per vertex)
        vbo[:, 3:6] = vbo[:, 0:3] / np.maximum(np.linalg.norm(vbo[:, 0:3], axis=1, keepdims=True), 1e-9)
        vbo[:, 6:9] = rng.uniform(0, 1, size=(vbo.shape[0], 3))
    else:              # cube, possibly with poisoned input
        P, I = ar.geometry.cube(float(rng.uniform(0.2, 4.0)))
        vbo, idx = ar.host.mesh_to_vbo(P, I)
        vbo = vbo.reshape(-1, 9).copy()
        if rng.random() < 0.5:
            vbo[rng.integers(0, vbo.shape[0]), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf, 1e30])
        idx = idx.copy()
        if rng.random() < 0.3:
            idx[rng.integers(0, idx.size)] = 0xFFFFFFF0
    cols, rows = int(rng.choice([1, 3, 37, 64, 65, 200, 333, 640])), int(rng.choice([1, 2, 31, 32, 33, 100, 257]))
    cam = {"pos": tuple(rng.uniform(-6, 6, size=3) * rng.choice([0.0, 0.3, 1.0])), "yaw": float(rng.uniform(-3.2, 3.2)),
           "pitch": float(rng.uniform(-1.5, 1.5))}
    proj, view = ar.host.camera_matrices(cam, cols, rows, float(rng.choice([0.5, 1.0])))
    npl = int(rng.choice([0, 0, 1, 4, 8]))
    scene = {"lights": {"points": [{"p": list(rng.uniform(-5, 5, size=3)), "color": list(rng.uniform(0, 1, size=3)),
                                    "intensity": float(rng.uniform(0, 30))} for _ in range(npl)]}} if npl else {}
    ramp = str(rng.choice(["@%#*+=-:. ", " .:-=+*#%@", "ab", "x", "".join(chr(33 + k) for k in range(90))]))
    mode = (bool(rng.random() < 0.7), int(rng.integers(1, 4)), int(rng.integers(1, 30)))
    T = idx.size // 3
    uv8 = atlas = None
    if rng.random() < 0.35:
        W, H = int(rng.choice([1, 2, 5, 32, 100])), int(rng.choice([1, 2, 7, 32]))
        uv8 = np.zeros((T, 8), dtype=np.uint16)
        uv8[:, :6] = rng.integers(0, max(W, H) + 3, size=(T, 6))
        uv8[:, 6] = rng.integers(0, 2, size=T)
        if rng.random() < 0.3:
            uv8[rng.integers(0, T), :6] = 65535
        atlas = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
        atlas[..., 3] = rng.choice(np.array([0, 1, 1, 32, 90, 126, 2, 31, 127, 255], dtype=np.uint8), size=(H, W))
    indexed = None
    if kind <= 2 and rng.random() < 0.4:       # hand the soup over as an indexed list and let the GPU build the vertex buffer
        indexed = (P, I, C)
    return {"uv8": uv8, "atlas": atlas, "indexed": indexed,
            "vbo": np.ascontiguousarray(vbo, dtype=np.float32), "idx": np.ascontiguousarray(idx, dtype=np.uint32), "proj": proj,
            "view": view, "lights": ar.host.pack_lights(scene), "cols": cols, "rows": rows, "ramp": ramp, "mode": mode,
            "path": int(rng.integers(0, 2)), "pool": int(rng.choice([0, 0, 0, 64, 5000])), "overlap": int(rng.integers(0, 3))}


def rt_case(rng):
    """Random ray-backend scene in the uniform layout, camera anywhere (also inside spheres)."""
    n_s, n_p, n_t, n_m = int(rng.choice([0, 1, 5, 32, 60])), int(rng.choice([0, 1, 3])), int(rng.choice([0, 2, 20, 64, 150])), int(rng.choice([0, 1, 6, 64]))
    sph = np.concatenate([rng.uniform(-6, 6, size=(n_s, 3)), rng.choice([0.0, 0.3, 1.0, 4.0], size=(n_s, 1))], axis=1)
    tri = rng.uniform(-7, 7, size=(n_t, 1, 3)) + rng.uniform(-2, 2, size=(n_t, 3, 3)) * rng.choice([0.0, 0.3, 1.0, 6.0], size=(n_t, 1, 1))
    pl = rng.uniform(-1, 1, size=(n_p, 4)); pl[:, 3] = rng.uniform(0, 9, size=n_p)
    n_pl, n_dl = int(rng.integers(0, 9)), int(rng.integers(0, 3))
    return {"uNumS": n_s, "uSpheres": sph, "uSm": rng.integers(-1, n_m + 2, size=n_s), "uNumP": n_p, "uPlanes": pl, "uPm": rng.integers(-1, n_m + 2, size=n_p),
            "uNumT": n_t, "uTa": tri[:, 0], "uTb": tri[:, 1], "uTc": tri[:, 2], "uTm": rng.integers(-1, n_m + 2, size=n_t),
            "uNumMats": n_m, "uMatAlbedo": rng.uniform(0, 1.2, size=(n_m, 3)), "uMatReflective": (rng.uniform(size=n_m) < 0.4).astype(np.int32),
            "uNumPL": n_pl, "uPLPos": rng.uniform(-8, 8, size=(n_pl, 3)), "uPLCol": rng.uniform(0, 25, size=(n_pl, 3)),
            "uNumDL": n_dl, "uDLDir": rng.uniform(-1, 1, size=(n_dl, 3)) * rng.choice([0.0, 1.0, 1.0]), "uDLCol": rng.uniform(0, 1.5, size=(n_dl, 3)),
            "uEnv": rng.uniform(0, 1.3, size=3)}


def rt_check(R, rng, n):
    u = rt_case(rng)
    cols, rows = int(rng.choice([1, 3, 37, 64, 200, 333])), int(rng.choice([1, 2, 31, 33, 100]))
    cam = (tuple(rng.uniform(-6, 6, size=3) * rng.choice([0.0, 0.3, 1.0])), float(rng.uniform(-3.2, 3.2)),
           float(rng.choice([rng.uniform(-1.5, 1.5), np.pi / 2, -np.pi / 2])), float(rng.uniform(0.3, 2.6)), float(rng.choice([0.5, 1.0, 0.4817])))
    mode = (bool(rng.random() < 0.7), int(rng.integers(1, 4)), int(rng.integers(1, 30)))
    ramp = "@%#*+=-:. "
    R.set_strip(0, 0); R.set_ascii(ramp, *mode); R.rt_set_scene(u)
    out = R.rt_render(cols, rows, *cam)
    want = orc.rt_render(u, cols, rows, *cam)
    ok = np.array_equal(out["rgba"], want) and np.array_equal(out["ascii"], orc.quantize(want, cols, rows, ramp, int(mode[0]), mode[1], mode[2]))
    if not ok:
        print("MISMATCH ray case %d: %d spheres %d planes %d tris grid %dx%d: rgba %d" % (
            n, u["uNumS"], u["uNumP"], u["uNumT"], cols, rows, np.count_nonzero(out["rgba"] != want)), flush=True)
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
    R = ar.Renderer(0)
    bad, t0 = 0, time.time()
    tot_tris = tot_cov = tot_cells = 0
    for n in range(cases):
        if n % 5 == 4:
            bad += 0 if rt_check(R, rng, n) else 1
            continue
        c = make_case(rng)
        R.set_option("path", c["path"]); R.set_option("pool_records", c["pool"]); R.set_option("overlap", min(c["overlap"], 1))
        R.set_option("tile_rows", (0, 8, 16, 32, 0)[n % 5])      # the tile height: chosen by the library, or each fixed one
        R.set_option("work_classes", 0 if n % 7 == 3 else 1)      # tile lists in entry order now and then
        R.set_option("fold_scan", 0 if n % 4 == 2 else 1)         # the scan kernel also where the frame would do without
        R.set_option("smoothing_kernel", (0, 3, 1, 3, 2, 3)[n % 6])  # each smoothing kernel where its parameters allow it
        if c["indexed"] is not None:
            R.set_mesh_indexed(*c["indexed"])
        else:
            R.set_mesh(c["vbo"], c["idx"])
        R.set_mesh_uv(c["uv8"]); R.set_texture(c["atlas"])
        R.set_camera(c["proj"], c["view"]); R.set_lights(c["lights"])
        # every third case puts the mesh under a model matrix: turned, scaled, moved a little (its own generator, so that
        # the other parameters of a case do not depend on it)
        model = None
        if n % 3 == 1:
            mrng = np.random.default_rng(1000003 * n + 17)
            model = ar.geometry.spin_matrix(float(mrng.uniform(-3.0, 3.0)), axis=tuple(mrng.normal(size=3) + 1e-3),
                                            translate=tuple(mrng.uniform(-0.3, 0.3, size=3)), scale=float(mrng.uniform(0.6, 1.6)))
        R.set_model(model)
        R.set_ascii(c["ramp"], *c["mode"])
        rows = c["rows"]
        strip = (0, 0)
        if rows >= 8 and rng.random() < 0.3:
            a = int(rng.integers(0, rows - 1)); strip = (a, int(rng.integers(a + 1, rows + 1)))
        R.set_strip(*strip)
        reps = int(rng.integers(1, 4))
        for _ in range(reps - 1):
            R.render_async(c["cols"], rows)
        out = R.render(c["cols"], rows, want_ids=True)
        rgba, depth, ids = orc.render(c["vbo"], c["idx"], c["proj"], c["view"], c["lights"], c["cols"], rows, uv8=c["uv8"], atlas=c["atlas"], model=model)
        asc = orc.quantize(rgba, c["cols"], rows, c["ramp"], int(c["mode"][0]), c["mode"][1], c["mode"][2])
        r0, r1 = strip if strip[1] > strip[0] else (0, rows)
        h0, h1 = max(0, r0 - 3), min(rows, r1 + 3)      # rows a strip renders at least (with smoothing halo)
        sl = slice(r0, r1)
        ok = (np.array_equal(out["ascii"][sl], asc[sl]) and np.array_equal(out["rgba"][sl], rgba[sl]) and
              np.array_equal(out["depth"][sl].view(np.uint32), depth[sl].view(np.uint32)) and np.array_equal(out["ids"][sl], ids[sl]))
        if ok and strip == (0, 0) and model is None and n % 2 == 0:
            # the same frame several times in one launch chain (the kernels take the frame from the grid's z index; chains
            # of four and more go through k_binstart)
            nb = 5 if n % 4 == 0 else 2
            R.render_batch_async(c["cols"], rows, R.pack_cameras([(c["proj"], c["view"])] * nb))
            for f in range(nb):
                got = R.read_batch(f)
                ok = ok and np.array_equal(got["ascii"], asc) and np.array_equal(got["rgba"], rgba)
        tot_tris += c["idx"].size // 3; tot_cov += int(np.count_nonzero(ids != 0xFFFFFFFF)); tot_cells += c["cols"] * rows
        if not ok:
            bad += 1
            print("MISMATCH case %d: tris %d grid %dx%d path %d pool %d overlap %d strip %s mode %s textured %s indexed %s: ids %d depth %d rgba %d ascii %d" % (
                n, c["idx"].size // 3, c["cols"], rows, c["path"], c["pool"], c["overlap"], strip, c["mode"], c["atlas"] is not None and c["atlas"].shape, c["indexed"] is not None,
                np.count_nonzero(out["ids"][sl] != ids[sl]), np.count_nonzero(out["depth"][sl].view(np.uint32) != depth[sl].view(np.uint32)),
                np.count_nonzero(out["rgba"][sl] != rgba[sl]), np.count_nonzero(out["ascii"][sl] != asc[sl])), flush=True)
        if n % 25 == 24:
            print("%d cases, %d mismatches, %.0f s" % (n + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d mismatches; %d triangles, %d of %d cells covered in total" % (cases, bad, tot_tris, tot_cov, tot_cells))
    sys.exit(1 if bad else 0)


main()
