"""Atlas texturing on the CPU side: the host's texel-coordinate rows against what the reference's path
tracer uploads (tests/golden/ref_pathtrace_uv.json, written by tests/golden/gen_pathtrace_uv.mjs), and the
oracle's texel rule against an image computed independently in numpy.

The reference applies this rule in its path tracer's GLSL only (shader_utils.js:94-132,
pathtrace_shader.js:141-152); no output of it exists in the reference tree, so the raster side of it is
PARITY UNPINNED and the oracle is its definition - pinned here to the written rule, not to a recording."""
import base64
import json
import os

import numpy as np
import pytest


def _u32(b):
    return np.frombuffer(base64.b64decode(b), dtype=np.uint32)


@pytest.fixture(scope="module")
def ref(golden_dir):
    with open(os.path.join(golden_dir, "ref_pathtrace_uv.json")) as fh:
        return json.load(fh)


def test_uv_rows_match_what_the_reference_uploads(pkg, ref):
    for case in ref["cases"]:
        scene, U = case["scene"], case["uniforms"]
        G = scene["geometry"]
        vbo, idx, uv8 = pkg.host.build_geometry(scene, textured=True)
        plain_vbo, plain_idx = pkg.host.build_geometry(scene)
        n_t, n_q = U["uNumTris"], U["uNumQuads"]
        assert n_t == len(G["tris"]) and n_q == len(G["quads"]), case["name"]
        assert idx.size // 3 == plain_idx.size // 3 + 4 * n_q
        # the reference-shaped part of the mesh is untouched by the extension
        assert np.array_equal(idx[:plain_idx.size], plain_idx)
        assert np.array_equal(vbo[:plain_vbo.shape[0], :6], plain_vbo[:, :6])
        # triangles come first (raster.js:171-175): row t = (uvA, uvB, uvC), always sampled
        if n_t:
            A, B, C = (_u32(U["uTriUV%s[0]" % k]).reshape(-1, 2)[:n_t] for k in "ABC")
            want = np.concatenate([A, B, C], axis=1)
            assert np.array_equal(uv8[:n_t, :6].astype(np.uint32), want), case["name"]
            assert np.all(uv8[:n_t, 6] == 1)
        # planes and spheres carry no coordinates
        first_quad = plain_idx.size // 3
        assert not uv8[n_t:first_quad].any()
        # quads: four triangles each, halves (0,1,2) and (0,2,3) and their mirror images
        Q = [_u32(U["uQuadUV%d[0]" % k]).reshape(-1, 2)[:n_q] for k in range(4)]
        for q in range(n_q):
            u = [tuple(int(x) for x in Q[k][q]) for k in range(4)]
            rows = uv8[first_quad + 4 * q: first_quad + 4 * q + 4]
            s1 = int(any(u[0] + u[1] + u[2]))
            s2 = int(any(u[0] + u[2] + u[3]))
            want = [u[0] + u[1] + u[2] + (s1, 0), u[0] + u[2] + u[3] + (s2, 0),
                    u[0] + u[2] + u[1] + (s1, 0), u[0] + u[3] + u[2] + (s2, 0)]
            assert rows.tolist() == [list(w) for w in want], "%s quad %d" % (case["name"], q)
        # the atlas goes to the texture unit byte for byte; without one the size uniform says 1 x 1 (off)
        if case["atlas_rgba"] is None:
            assert U["uAtlasSize"] == [1, 1] and len(case["tex_uploads"]) == 1
        else:
            up = case["tex_uploads"][-1]
            assert [up["width"], up["height"]] == U["uAtlasSize"]
            assert up["data"] == case["atlas_rgba"] and up["format"] == "RGBA" and up["type"] == "UNSIGNED_BYTE"


def test_quads_are_two_sided_and_coloured_from_the_material_table(pkg, ref):
    scene = ref["cases"][0]["scene"]
    vbo, idx, uv8 = pkg.host.build_geometry(scene, textured=True)
    first = pkg.host.build_geometry(scene)[1].size // 3
    tri = vbo[idx.reshape(-1, 3)]                       # [T, 3, 9]
    for q in range(len(scene["geometry"]["quads"])):
        t = tri[first + 4 * q: first + 4 * q + 4]
        assert np.allclose(t[0, 0, 3:6], -t[2, 0, 3:6]) and np.allclose(t[1, 0, 3:6], -t[3, 0, 3:6])   # mirrored normals
        assert np.array_equal(t[0, :, :3], t[2, [0, 2, 1], :3])                                          # same corners
    white = scene["materials"]["table"]["1"]["albedo"]
    assert np.allclose(tri[first, 0, 6:9], white)       # the floor is WHITE (js/scene.js:16)
    assert np.allclose(tri[0, 0, 6:9], scene["materials"]["table"]["6"]["albedo"])   # first sphere: GLASS


def _full_view_quad(W, H, colour):
    """Two triangles filling the viewport at depth 0.5 under identity matrices; texel coordinates
    (0,0) bottom-left to (W,H) top-right."""
    P = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32)
    vbo = np.zeros((4, 9), dtype=np.float32)
    vbo[:, :3] = P
    vbo[:, 3:6] = (0, 0, 1)
    vbo[:, 6:9] = colour
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    uv8 = np.array([[0, 0, W, 0, W, H, 1, 0], [0, 0, W, H, 0, H, 1, 0]], dtype=np.uint16)
    return vbo, idx, uv8


def _class_atlas(W, H, seed):
    rng = np.random.default_rng(seed)
    atlas = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    alpha = rng.choice(np.array([0, 1, 1, 32, 65, 126, 2, 31, 127, 255], dtype=np.uint8), size=(H, W))
    atlas[..., 3] = alpha
    return atlas


WHITE_AMBIENT = {"ambient": (1, 1, 1), "dir": (0, 0, -1), "dir_color": (0, 0, 0)}


def test_oracle_texel_rule_against_an_independent_image(oracle):
    """Cell (r, i) of a 2W x 2H grid sees texel coordinates ((i+0.5)/2, (j+0.5)/2), j = rows-1-r: nearest
    texel floor(x + 0.5) = floor(i/2 + 0.75); the texel is read from uploaded row H-1-ty."""
    W, H = 13, 7
    cols, rows = 2 * W, 2 * H
    colour = (0.2, 0.4, 0.6)
    vbo, idx, uv8 = _full_view_quad(W, H, colour)
    atlas = _class_atlas(W, H, 3)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    rgba, depth, ids = oracle.render(vbo, idx, ident, ident, WHITE_AMBIENT, cols, rows, uv8=uv8, atlas=atlas)
    want = np.zeros((rows, cols, 4), dtype=np.uint8)
    classes = set()
    for r in range(rows):
        j = rows - 1 - r
        for i in range(cols):
            tx, ty = (2 * i + 3) // 4, (2 * j + 3) // 4       # floor(i/2 + 0.75)
            px = (51, 102, 153, 255)                          # round(255 * colour)
            if tx < W and ty < H:
                t = atlas[H - 1 - ty, tx]
                if 32 <= t[3] <= 126:
                    px = tuple(t); classes.add("glyph")
                elif t[3] == 1:
                    px = (t[0], t[1], t[2], 255); classes.add("solid")
                else:
                    classes.add("none")
            else:
                classes.add("outside")
            want[r, i] = px
    assert classes == {"glyph", "solid", "none", "outside"}
    assert np.array_equal(rgba, want)
    assert np.all(depth == 0.5) and np.all(ids != 0xFFFFFFFF)
    # the override travels on into the ASCII stage: glyph cells come out as their code
    codes = oracle.quantize(rgba, cols, rows, "@%#*+=-:. ", 0)
    glyph = (want[..., 3] >= 32) & (want[..., 3] <= 126)
    assert np.array_equal(codes[glyph], want[..., 3][glyph])


def test_oracle_texturing_off_cases(oracle):
    W, H = 8, 8
    vbo, idx, uv8 = _full_view_quad(W, H, (0.5, 0.5, 0.5))
    atlas = _class_atlas(W, H, 5)
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    plain = oracle.render(vbo, idx, ident, ident, WHITE_AMBIENT, 16, 16)[0]
    # an atlas one texel wide or high switches texturing off (shader_utils.js:94)
    for shape in ((1, W), (H, 1), (1, 1)):
        got = oracle.render(vbo, idx, ident, ident, WHITE_AMBIENT, 16, 16, uv8=uv8, atlas=atlas[:shape[0], :shape[1]])[0]
        assert np.array_equal(got, plain)
    # triangles whose sample bit is clear stay untextured
    off = uv8.copy(); off[:, 6] = 0
    assert np.array_equal(oracle.render(vbo, idx, ident, ident, WHITE_AMBIENT, 16, 16, uv8=off, atlas=atlas)[0], plain)
    # and with the bit set the image does change
    assert not np.array_equal(oracle.render(vbo, idx, ident, ident, WHITE_AMBIENT, 16, 16, uv8=uv8, atlas=atlas)[0], plain)


def test_oracle_solid_texel_is_lit_like_a_vertex_colour(oracle):
    """One solid texel stretched over the view, a slanted light: the image equals the untextured render
    of the same quad carrying the texel's rgb / 255 as vertex colour."""
    vbo, idx, uv8 = _full_view_quad(1, 1, (0.9, 0.1, 0.1))
    uv8[:, :6] = 1                                       # every vertex at texel (1, 1)
    atlas = np.zeros((2, 2, 4), dtype=np.uint8)
    atlas[0, 1] = (200, 100, 50, 1)                      # texel (1, 1) lives in uploaded row H-1-1 = 0
    lights = {"ambient": (0.1, 0.2, 0.3), "dir": (0.3, -0.2, -1.0), "dir_color": (0.9, 0.8, 0.7),
              "pl_pos": [(0.5, 0.5, 1.0)], "pl_col": [(0.5, 0.5, 0.5)]}
    ident = np.eye(4, dtype=np.float32).reshape(-1)
    got = oracle.render(vbo, idx, ident, ident, lights, 12, 9, uv8=uv8, atlas=atlas)[0]
    vbo2 = vbo.copy()
    vbo2[:, 6:9] = np.array([200, 100, 50], dtype=np.float32) / np.float32(255)
    want = oracle.render(vbo2, idx, ident, ident, lights, 12, 9)[0]
    assert np.array_equal(got, want) and len(np.unique(got.reshape(-1, 4), axis=0)) > 3


def test_default_scene_poster_shows_its_glyphs(pkg, oracle, ref):
    """The reference's default scene with a camera facing the poster (js/scene.js:28-48): glyph texels
    arrive as alpha overrides inside the poster's outline, and the clipped room walls still render."""
    case = ref["cases"][0]
    scene = case["scene"]
    atlas = np.frombuffer(base64.b64decode(case["atlas_rgba"]), dtype=np.uint8).reshape(32, 32, 4)
    vbo, idx, uv8 = pkg.host.build_geometry(scene, textured=True)
    cols, rows = 120, 50
    proj, view = pkg.host.camera_matrices({"pos": (0.0, 2.5, 6.0), "yaw": -np.pi / 2, "pitch": 0.0}, cols, rows, 0.5)
    lights = pkg.host.pack_lights(scene)
    lights["ambient"] = (0.4, 0.4, 0.4)
    rgba, depth, ids = oracle.render(vbo, idx, proj, view, lights, cols, rows, uv8=uv8, atlas=atlas)
    a = rgba[..., 3]
    glyph = (a >= 32) & (a <= 126)
    assert glyph.sum() > 100
    first_quad = pkg.host.build_geometry(scene)[1].size // 3
    poster = first_quad + 4 * 6                           # seventh quad of js/scene.js
    assert set(np.unique(ids[glyph] >> 1).tolist()) <= set(range(poster, poster + 4))
    assert np.all(ids != 0xFFFFFFFF)                      # inside a closed room every cell is covered
    codes = oracle.quantize(rgba, cols, rows, "@%#*+=-:. ", 1, 2, 12)
    assert np.array_equal(codes[glyph], a[glyph])         # overrides survive the smoothing pass
