"""The raster half of the CPU oracle (orc_render: vertex transform, near clip, culling, coverage, depth test,
perspective-correct varyings, the fragment shader, the row flip) checked on the CPU against

  (a) an independent binary64 renderer written from the GL rules, not from the oracle (tests/independent_raster.py),
  (b) answers worked out by hand: an axis-aligned rectangle facing the camera (exact cell rectangle, depth from the
      projection matrix, colour from the Lambert formula of raster_shader.js:42-62), culling of back faces and
      zero-area triangles, and "the earlier triangle wins an equal depth",
  (c) self-consistency frames committed under tests/golden/ (made by tests/golden/gen_self_consistency.py).

The reference runs this stage inside the browser's WebGL implementation and holds no expected output for it, so
parity with the reference itself stays unpinned; these tests make sure the oracle is at least a correct rasterizer
by an independent account, and that it does not drift."""
import math
import os

import numpy as np
import pytest

from independent_raster import render as independent_render


def _compare(oracle, sc, min_covered, max_ambiguous=0.05):
    rgba, depth, ids = oracle.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], sc["cols"], sc["rows"],
                                     model=sc.get("model"))
    ind = independent_render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], sc["cols"], sc["rows"],
                             model=sc.get("model"))
    covered = ids != 0xFFFFFFFF
    clear = ~ind["ambiguous"]
    assert covered.sum() >= min_covered
    # coverage: identical wherever rounding cannot decide
    assert np.array_equal(covered[clear], ind["covered"][clear]), \
        "%s: %d cells differ in coverage" % (sc["name"], np.count_nonzero(covered[clear] != ind["covered"][clear]))
    assert ind["ambiguous"].mean() < max_ambiguous
    both = covered & ind["covered"] & clear
    assert np.all(depth[~covered] == 1.0) and np.all(rgba[~covered] == (0, 0, 0, 255))
    if not both.any():
        return 0
    # depth within 1e-5 (the tolerance the build is held to), colour bytes within 1
    assert np.max(np.abs(depth[both].astype(np.float64) - ind["depth"][both])) <= 1e-5
    diff = np.abs(rgba[both].astype(np.int16) - ind["rgba"][both].astype(np.int16))
    assert diff.max() <= 1, "%s: colour bytes differ by %d" % (sc["name"], diff.max())
    # ... and a byte may only differ where binary64 itself sits next to a rounding boundary
    off = diff[..., :3].max(axis=-1) > 0
    assert np.all(ind["byte_margin"][both][off] < 2e-3)
    assert off.mean() < 0.02
    return both.sum()


def test_c1_cube_orbit_matches_the_independent_renderer(pkg, oracle):
    total = 0
    for k in range(60):
        total += _compare(oracle, pkg.scenes.cube_scene(angle=2 * math.pi * k / 60), 100)
    assert total > 60 * 100


def test_c2_icosphere_matches_the_independent_renderer(pkg, oracle):
    assert _compare(oracle, pkg.scenes.icosphere_scene(point_lights=3), 20_000) > 20_000


def test_smooth_normals_and_reference_default_scene(pkg, oracle):
    """The reference's default scene as its raster backend sees it: two UV spheres with smooth normals."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_raster_host.json")) as fh:
        ref = json.load(fh)
    done = seen = 0
    for case in ref["cases"]:
        scene = ref["scenes"][case["scene"]]
        vbo, idx = pkg.host.build_geometry(scene)
        if idx.size == 0 or idx.size > 3 * 4000:
            continue
        proj, view = pkg.host.camera_matrices(case["camera"], case["cols"], case["rows"], case["pixelAspect"])
        sc = {"name": case["scene"], "vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": pkg.host.pack_lights(scene),
              "cols": case["cols"], "rows": case["rows"]}
        # (the legacy scene stacks coplanar planes: every cell of theirs is a depth tie, which only the cells that
        # remain can speak about)
        seen += _compare(oracle, sc, 0, max_ambiguous=0.8)
        done += 1
    assert done >= 2 and seen > 500


def test_model_matrix_matches_the_independent_renderer(pkg, oracle):
    """The uModel definition of orc_render_model: the cube of C1 spun about a tilted axis under a fixed camera,
    and a lit icosphere scaled, turned and pushed towards the camera until the near plane cuts it."""
    total = 0
    for k in range(0, 60, 4):
        sc = pkg.scenes.cube_scene(angle=0.6)
        sc["model"] = pkg.geometry.spin_matrix(2 * math.pi * k / 60, axis=(0.3, 1.0, 0.2))
        total += _compare(oracle, sc, 100)
    assert total > 15 * 100
    sc = pkg.scenes.icosphere_scene(cols=200, rows=100, subdiv=3, point_lights=3)
    sc["model"] = pkg.geometry.spin_matrix(0.8, axis=(1.0, 0.2, -0.4), translate=(0.4, 0.1, -0.3), scale=1.7)
    _compare(oracle, sc, 3000)
    # triangles of every size around and through the camera, so that the near plane cuts moved geometry
    rng = np.random.default_rng(9)
    T = 700
    centre = rng.uniform(-6, 6, size=(T, 1, 3))
    P = (centre + rng.uniform(-1, 1, size=(T, 3, 3)) * rng.choice([0.05, 0.5, 4.0, 30.0], size=(T, 1, 3))).reshape(-1, 3)
    vbo, idx = pkg.host.mesh_to_vbo(P.astype(np.float32), np.arange(T * 3).reshape(-1, 3), rng.uniform(0, 1, size=(T, 3)))
    proj, view = pkg.host.camera_matrices({"pos": (0.1, 0.2, 0.3), "yaw": 0.7, "pitch": -0.2}, 160, 80, 0.5)
    lights = pkg.host.pack_lights({"lights": {"points": [{"p": [1, 2, 0], "color": [1, 1, 1], "intensity": 4}]}})
    soup = {"name": "soup under a model matrix", "vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": lights, "cols": 160, "rows": 80,
            "model": pkg.geometry.spin_matrix(0.5, axis=(0.2, 0.1, 1.0), translate=(0.3, -0.2, 0.1), scale=0.8)}
    _compare(oracle, soup, 160 * 80 // 2)
    ids = oracle.render(vbo, idx, proj, view, lights, 160, 80, model=soup["model"])[2]
    assert np.count_nonzero((ids != 0xFFFFFFFF) & ((ids & 1) == 1)) > 50        # second halves of clipped triangles are on screen
    # an identity model matrix changes nothing, bit for bit
    sc = pkg.scenes.icosphere_scene(cols=160, rows=60, subdiv=2, point_lights=2)
    a = oracle.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], 160, 60)
    b = oracle.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], 160, 60, model=np.eye(4, dtype=np.float32).reshape(-1))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_clip_soup_matches_the_independent_renderer(pkg, oracle):
    """Triangles of every size around and through the camera: the near clip, culling after clipping, huge cell ranges."""
    rng = np.random.default_rng(7)
    T = 1200
    centre = rng.uniform(-6, 6, size=(T, 1, 3))
    P = (centre + rng.uniform(-1, 1, size=(T, 3, 3)) * rng.choice([0.05, 0.5, 4.0, 30.0], size=(T, 1, 1))).reshape(-1, 3)
    vbo, idx = pkg.host.mesh_to_vbo(P.astype(np.float32), np.arange(T * 3).reshape(-1, 3), rng.uniform(0, 1, size=(T, 3)))
    lights = pkg.host.pack_lights({"lights": {"points": [{"p": [1, 2, 0], "color": [1, 1, 1], "intensity": 4}]}})
    for cols, rows, cam in [(200, 100, {"pos": (0.1, 0.2, 0.3), "yaw": 0.7, "pitch": -0.2}),
                            (97, 61, {"pos": (3, -2, 1), "yaw": -2.0, "pitch": 0.9})]:
        proj, view = pkg.host.camera_matrices(cam, cols, rows, 0.5)
        sc = {"name": "soup", "vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": lights, "cols": cols, "rows": rows}
        _compare(oracle, sc, cols * rows // 2)


# ---- (b) answers worked out by hand ---------------------------------------------------------------------------

def _rect_facing_camera(d, y0, y1, z0, z1, col, flip=False):
    """Two triangles spanning [y0, y1] x [z0, z1] in the plane x = d, counter-clockwise as seen from the origin (the
    camera looks along +X with screen right = +Z, up = +Y), normal (-1, 0, 0)."""
    quad = np.array([[d, y0, z0], [d, y0, z1], [d, y1, z1], [d, y1, z0]], dtype=np.float32)
    tris = [(0, 1, 2), (0, 2, 3)]
    if flip:
        tris = [(a, c, b) for a, b, c in tris]
    vbo = np.zeros((6, 9), dtype=np.float32)
    for t, tri in enumerate(tris):
        for k, v in enumerate(tri):
            vbo[3 * t + k] = np.concatenate([quad[v], [-1.0, 0.0, 0.0], col])
    return vbo, np.arange(6, dtype=np.uint32)


def _camera(pkg, cols, rows):
    return pkg.host.camera_matrices({"pos": (0.0, 0.0, 0.0), "yaw": 0.0, "pitch": 0.0}, cols, rows, 1.0)


def test_rectangle_facing_the_camera_by_hand(pkg, oracle):
    cols, rows, d = 64, 48, 4.0
    proj, view = _camera(pkg, cols, rows)
    f = 1.0 / math.tan(math.radians(40.0))                       # raster.js:297: the literal 80 degree fallback
    aspect = cols / rows
    # window position of a point (d, y, z): x_w = (z f / (aspect d) + 1) cols / 2, y_w = (y f / d + 1) rows / 2.
    # Choose the rectangle so that its edges fall at window x = 10.25 / 40.75 and y = 8.25 / 30.75: the covered cell
    # centres are then columns 10..40 and GL rows 8..30, i.e. output rows rows-1-30 .. rows-1-8.
    z_of = lambda xw: (xw * 2.0 / cols - 1.0) * aspect * d / f
    y_of = lambda yw: (yw * 2.0 / rows - 1.0) * d / f
    col = np.array([0.8, 0.5, 0.25], dtype=np.float32)
    vbo, idx = _rect_facing_camera(d, y_of(8.25), y_of(30.75), z_of(10.25), z_of(40.75), col)
    lights = pkg.host.pack_lights({})                            # raster.js:66-68 defaults
    rgba, depth, ids = oracle.render(vbo, idx, proj, view, lights, cols, rows)
    want = np.zeros((rows, cols), dtype=bool)
    want[rows - 1 - 30:rows - 1 - 8 + 1, 10:41] = True
    assert np.array_equal(ids != 0xFFFFFFFF, want)
    # depth: z_ndc = (P[10] (-d) + P[14]) / d with the projection's near 0.05 / far 100, window depth (z_ndc + 1) / 2
    near, far = 0.05, 100.0
    z_ndc = ((far + near) / (near - far) * (-d) + 2.0 * far * near / (near - far)) / d
    assert np.max(np.abs(depth[want].astype(np.float64) - (z_ndc + 1.0) / 2.0)) < 1e-6
    assert np.all(depth[~want] == 1.0)
    # colour: N = (-1, 0, 0); ambient + directional with the default light, no point lights
    amb, ldir, lcol = np.array([0.15, 0.18, 0.22]), np.array([0.25, -1.0, 0.15]), np.array([1.2, 1.15, 1.1])
    ndl = max(float(np.dot([-1.0, 0.0, 0.0], -ldir)), 0.0)        # = 0.25
    c = col.astype(np.float64) * amb + col.astype(np.float64) * lcol * ndl
    byte = np.floor(np.clip(c, 0, 1) * 255 + 0.5).astype(np.uint8)
    assert np.all(rgba[want][:, :3] == byte) and np.all(rgba[..., 3] == 255)
    assert np.all(rgba[~want][:, :3] == 0)
    # with a point light one unit in front of the rectangle's centre cell: the Lambert term by hand at that cell
    j, i = 19, 25
    centre = np.array([d, y_of(j + 0.5), z_of(i + 0.5)])
    lp = centre + np.array([-1.0, 0.0, 0.0])
    lights2 = pkg.host.pack_lights({"lights": {"points": [{"p": lp.tolist(), "color": [0.5, 1.0, 0.25], "intensity": 2.0}]}})
    rgba2, _d, _i = oracle.render(vbo, idx, proj, view, lights2, cols, rows)
    d2 = 1.0
    c2 = c + col.astype(np.float64) * np.array([1.0, 2.0, 0.5]) * (1.0 * (1.0 / (1.0 + 0.05 * d2)))
    byte2 = np.floor(np.clip(c2, 0, 1) * 255 + 0.5)
    assert np.all(np.abs(rgba2[rows - 1 - j, i, :3].astype(np.float64) - byte2) <= 1)


def test_back_faces_and_zero_area_are_culled(pkg, oracle):
    cols, rows = 40, 30
    proj, view = _camera(pkg, cols, rows)
    lights = pkg.host.pack_lights({})
    vbo, idx = _rect_facing_camera(3.0, -1.0, 1.0, -1.0, 1.0, np.array([1, 1, 1], dtype=np.float32), flip=True)
    _rgba, depth, ids = oracle.render(vbo, idx, proj, view, lights, cols, rows)
    assert np.all(ids == 0xFFFFFFFF) and np.all(depth == 1.0)
    # zero area: three collinear vertices, and a triangle seen exactly edge-on
    v = np.zeros((6, 9), dtype=np.float32)
    v[0, :3], v[1, :3], v[2, :3] = (3, -1, -1), (3, 0, 0), (3, 1, 1)
    v[3, :3], v[4, :3], v[5, :3] = (2, 0, -1), (4, 0, 1), (3, 0, 0.5)          # in the plane y = 0 through the eye
    v[:, 3:6] = (-1, 0, 0)
    v[:, 6:9] = 1
    _rgba, depth, ids = oracle.render(v, np.arange(6, dtype=np.uint32), proj, view, lights, cols, rows)
    assert np.all(ids == 0xFFFFFFFF)
    # behind the camera and beyond the far plane
    vbo, idx = _rect_facing_camera(-3.0, -1.0, 1.0, -1.0, 1.0, np.array([1, 1, 1], dtype=np.float32))
    assert np.all(oracle.render(vbo, idx, proj, view, lights, cols, rows)[2] == 0xFFFFFFFF)
    vbo, idx = _rect_facing_camera(150.0, -10.0, 10.0, -10.0, 10.0, np.array([1, 1, 1], dtype=np.float32))
    assert np.all(oracle.render(vbo, idx, proj, view, lights, cols, rows)[2] == 0xFFFFFFFF)


def test_equal_depth_goes_to_the_earlier_triangle_and_nearer_wins(pkg, oracle):
    cols, rows = 40, 30
    proj, view = _camera(pkg, cols, rows)
    lights = pkg.host.pack_lights({})
    red, green = np.array([1, 0, 0], dtype=np.float32), np.array([0, 1, 0], dtype=np.float32)
    a, ia = _rect_facing_camera(3.0, -1.0, 1.0, -1.0, 1.0, red)
    b, _ = _rect_facing_camera(3.0, -1.0, 1.0, -1.0, 1.0, green)
    vbo = np.concatenate([a, b]); idx = np.arange(12, dtype=np.uint32)
    rgba, _depth, ids = oracle.render(vbo, idx, proj, view, lights, cols, rows)
    hit = ids != 0xFFFFFFFF
    assert hit.sum() > 100
    assert np.all((ids[hit] >> 1) < 2)                            # the first copy's two triangles own every cell
    assert np.all(rgba[hit][:, 1] == 0) and np.all(rgba[hit][:, 0] > 0)
    # same pair submitted in the other order: green now
    rgba, _depth, ids = oracle.render(np.concatenate([b, a]), idx, proj, view, lights, cols, rows)
    assert np.all(rgba[hit][:, 0] == 0) and np.all(rgba[hit][:, 1] > 0)
    # a nearer rectangle submitted later still wins (depth LESS, not submission order)
    near_rect, _ = _rect_facing_camera(2.0, -0.3, 0.3, -0.3, 0.3, green)
    rgba, depth, ids = oracle.render(np.concatenate([a, near_rect]), idx, proj, view, lights, cols, rows)
    mid = (rows // 2, cols // 2)
    assert ids[mid] >> 1 >= 2 and rgba[mid][1] > 0 and rgba[mid][0] == 0
    assert depth[mid] < depth[rows // 2, 2 + cols // 2 + 8] or ids[rows // 2, 2 + cols // 2 + 8] == 0xFFFFFFFF


def test_shared_edges_are_owned_exactly_once_on_the_cpu(pkg, oracle):
    """A fan of triangles around a vertex that sits exactly on a cell centre, and a grid of rectangles whose
    shared edges run exactly through cell centres: every cell is covered by exactly one triangle - counted by
    rendering each triangle alone and adding up."""
    cols, rows, d = 32, 32, 2.0
    proj, view = _camera(pkg, cols, rows)
    f = 1.0 / math.tan(math.radians(40.0))
    z_of = lambda xw: (xw * 2.0 / cols - 1.0) * d / f
    y_of = lambda yw: (yw * 2.0 / rows - 1.0) * d / f
    lights = pkg.host.pack_lights({})
    # 2 x 2 rectangles meeting at window point (16.5, 16.5) - a cell centre - with edges along x = 16.5 and y = 16.5
    xs, ys = [4.5, 16.5, 28.5], [4.5, 16.5, 28.5]
    parts = []
    for a in range(2):
        for b in range(2):
            v, _ = _rect_facing_camera(d, y_of(ys[a]), y_of(ys[a + 1]), z_of(xs[b]), z_of(xs[b + 1]), np.array([1, 1, 1], dtype=np.float32))
            parts.append(v)
    vbo = np.concatenate(parts)
    count = np.zeros((rows, cols), dtype=np.int32)
    for t in range(vbo.shape[0] // 3):
        ids = oracle.render(vbo[3 * t:3 * t + 3], np.arange(3, dtype=np.uint32), proj, view, lights, cols, rows)[2]
        count += ids != 0xFFFFFFFF
    all_ids = oracle.render(vbo, np.arange(vbo.shape[0], dtype=np.uint32), proj, view, lights, cols, rows)[2]
    inside = all_ids != 0xFFFFFFFF
    # binary32 vertex positions put the shared edges within rounding of the cell centres: whichever side they fall,
    # no cell may be claimed twice and none inside the outline may be dropped
    assert np.all(count[inside] == 1) and np.all(count[~inside] == 0)
    assert 23 * 23 <= inside.sum() <= 25 * 25


# ---- (c) committed self-consistency frames --------------------------------------------------------------------

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "self_consistency_frames.npz")


def golden_cases(pkg):
    """(name, scene) of every committed frame, in the order gen_self_consistency.py wrote them: C1 (sixty orbit angles),
    C2 without and with point lights, and one frame of each regime most likely to drift together with the kernels - a
    soup of triangles through the near plane (200 x 100), the cube under a model matrix, the reference's own default
    scene (two smooth-shaded spheres, 96 x 36), that scene with its quads and atlas textured, and 64 k small triangles
    shaped like C3 on a 320 x 180 grid."""
    import base64
    import json
    cases = [("c1_%02d" % k, pkg.scenes.cube_scene(angle=2 * math.pi * k / 60)) for k in range(60)]
    cases.append(("c2", pkg.scenes.icosphere_scene()))
    cases.append(("c2_lights", pkg.scenes.icosphere_scene(angle=1.9, point_lights=4)))
    host, golden = pkg.host, os.path.join(os.path.dirname(__file__), "golden")
    # near-plane clipping: big and small triangles around and through the camera, one point light
    rng = np.random.default_rng(7)
    T = 3000
    centre = rng.uniform(-6, 6, size=(T, 1, 3))
    P = (centre + rng.uniform(-1, 1, size=(T, 3, 3)) * rng.choice([0.05, 0.5, 4.0, 30.0], size=(T, 1, 1))).reshape(-1, 3)
    vbo, idx = host.mesh_to_vbo(P.astype(np.float32), np.arange(T * 3).reshape(-1, 3), rng.uniform(0, 1, size=(T, 3)))
    proj, view = host.camera_matrices({"pos": (0.1, 0.2, 0.3), "yaw": 0.7, "pitch": -0.2}, 200, 100, 0.5)
    lights = host.pack_lights({"lights": {"points": [{"p": [1, 2, 0], "color": [1, 1, 1], "intensity": 4}]}})
    cases.append(("clip_soup", {"name": "clip soup", "vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": lights, "cols": 200, "rows": 100}))
    # a model matrix (rotation, translation, scale) on C1's cube
    cases.append(("model_cube", dict(pkg.scenes.cube_scene(angle=0.6), model=pkg.geometry.spin_matrix(0.9, axis=(0.3, 1.0, 0.2), translate=(0.2, -0.1, 0.3), scale=1.3))))
    # the reference's default scene as its rasterizer draws it: two spheres with smooth normals
    with open(os.path.join(golden, "ref_raster_host.json")) as fh:
        ref = json.load(fh)
    case = next(c for c in ref["cases"] if c["scene"] == "default_scene" and c["cols"] == 96 and c["rows"] == 36)
    scene = ref["scenes"]["default_scene"]
    vbo, idx = host.build_geometry(scene)
    proj, view = host.camera_matrices(case["camera"], 96, 36, case["pixelAspect"])
    cases.append(("default_scene", {"name": "default scene", "vbo": vbo, "idx": idx, "proj": proj, "view": view, "lights": host.pack_lights(scene), "cols": 96, "rows": 36}))
    # the same scene with its quads and the 32 x 32 atlas (ASCII overrides in the alpha bytes), facing the poster
    with open(os.path.join(golden, "ref_pathtrace_uv.json")) as fh:
        tcase = next(c for c in json.load(fh)["cases"] if c["name"] == "default_scene_atlas32")
    tscene = tcase["scene"]
    atlas = np.frombuffer(base64.b64decode(tcase["atlas_rgba"]), dtype=np.uint8).reshape(tscene["atlas"]["height"], tscene["atlas"]["width"], 4)
    vbo, idx, uv8 = host.build_geometry(tscene, textured=True)
    proj, view = host.camera_matrices({"pos": (0.0, 2.5, 6.0), "yaw": -math.pi / 2, "pitch": 0.0}, 120, 50, 0.5)
    cases.append(("textured", {"name": "textured default scene", "vbo": vbo, "idx": idx, "uv8": uv8, "atlas": atlas, "proj": proj, "view": view,
                               "lights": host.pack_lights(tscene), "cols": 120, "rows": 50}))
    # dense small triangles: C3's distribution at 64 k triangles on a 320 x 180 grid (about three cells each)
    cases.append(("c3_small", pkg.scenes.random_scene(65536, 320, 180, mesh_grid=(320, 180))))
    return cases


def _oracle_frame(pkg, oracle, sc):
    rgba, depth, _ids = oracle.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], sc["cols"], sc["rows"], want_ids=False,
                                      uv8=sc.get("uv8"), atlas=sc.get("atlas"), model=sc.get("model"))
    return rgba, depth, oracle.quantize(rgba, sc["cols"], sc["rows"], pkg.scenes.DEFAULT_RAMP, 1, 2, 12)


def test_oracle_reproduces_the_committed_frames(pkg, oracle):
    z = np.load(GOLDEN)
    names = set()
    for name, sc in golden_cases(pkg):
        rgba, depth, asc = _oracle_frame(pkg, oracle, sc)
        assert np.array_equal(asc, z[name + "_ascii"]), name
        assert np.array_equal(depth.view(np.uint32), z[name + "_depth"].view(np.uint32)), name
        assert np.array_equal(rgba, z[name + "_rgba"]), name
        names.update(name + suffix for suffix in ("_ascii", "_depth", "_rgba"))
    assert names == set(z.files)
    # the frames show what their names say
    assert np.count_nonzero(z["clip_soup_depth"] < 1.0) > 5000 and np.count_nonzero(z["c3_small_depth"] < 1.0) > 10000
    a = z["textured_rgba"][..., 3]
    assert np.count_nonzero((a >= 32) & (a <= 126)) > 100 and np.all(z["default_scene_rgba"][..., 3] == 255)


@pytest.mark.gpu
def test_hip_path_reproduces_the_committed_frames(pkg):
    """The frames frozen under tests/golden/ straight against the HIP path: no oracle in the loop."""
    z = np.load(GOLDEN)
    for path in (0, 1):
        R = pkg.Renderer(0)
        try:
            R.set_option("path", path)
            for name, sc in golden_cases(pkg):
                pkg.scenes.upload(R, sc)
                out = R.render(sc["cols"], sc["rows"])
                assert out["rgba"].shape == z[name + "_rgba"].shape, (path, name)
                assert np.array_equal(out["ascii"], z[name + "_ascii"]), (path, name)
                assert np.array_equal(out["depth"].view(np.uint32), z[name + "_depth"].view(np.uint32)), (path, name)
                assert np.array_equal(out["rgba"], z[name + "_rgba"]), (path, name)
        finally:
            R.close()
