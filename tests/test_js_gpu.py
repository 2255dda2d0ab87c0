"""The JavaScript host end to end on the GPU: Node -> router -> HIP backend -> N-API addon -> HIP
kernels, compared bit for bit with the oracle fed by the Python host mirror."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_node_harness_frames_match_oracle(pkg, oracle, tmp_path, rt_uniforms_of):
    node = shutil.which("node")
    assert node, "node is required for the JavaScript host"
    r = subprocess.run([node, os.path.join(ROOT, "tests", "js", "gpu_frame.mjs"), str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    meta = json.load(open(tmp_path / "meta.json"))
    assert meta["row5"] == "3.14159 fps"
    ramp = "@%#*+=-:. "
    meshes = {
        "cube": (pkg.geometry.cube(0.5), None),
        "cube_mesh": (pkg.geometry.cube(0.5), None),
        "cube_small": (pkg.geometry.cube(0.5), None),
        "ico": (pkg.geometry.icosphere(4, 1.0),
                {"lights": {"points": [{"p": [2, 1.5, 0], "color": [1, 0.6, 0.5], "intensity": 3},
                                       {"p": [-1, 1.5, 1.7], "color": [1, 0.7, 0.5], "intensity": 3}]}}),
    }
    for fr in meta["frames"]:
        (P, I), scene = meshes[fr["name"]]
        vbo, idx = pkg.host.mesh_to_vbo(P, I)
        cols, rows = fr["cols"], fr["rows"]
        proj, view = pkg.host.camera_matrices(fr["cam"], cols, rows, fr["pixelAspect"])
        lights = pkg.host.pack_lights(scene or {})
        if fr["name"] == "cube":          # SceneBuilder scenes: ambient = env colour * 0 (raster.js:309-311)
            lights["ambient"] = np.zeros(3, dtype=np.float32)
        rgba, depth, _ = oracle.render(vbo, idx, proj, view, lights, cols, rows)
        asc = oracle.quantize(rgba, cols, rows, ramp, 1, 2, 12)
        got_rgba = np.fromfile(tmp_path / (fr["name"] + ".rgba"), dtype=np.uint8).reshape(rows, cols, 4)
        got_depth = np.fromfile(tmp_path / (fr["name"] + ".depth"), dtype=np.float32).reshape(rows, cols)
        got_ascii = np.fromfile(tmp_path / (fr["name"] + ".ascii"), dtype=np.uint8).reshape(rows, cols)
        assert np.array_equal(got_rgba, rgba), fr["name"]
        assert np.array_equal(got_depth.view(np.uint32), depth.view(np.uint32)), fr["name"]
        assert np.array_equal(got_ascii, asc), fr["name"]
        assert np.count_nonzero(rgba[:, :, :3]) > 30 and np.count_nonzero(depth < 1.0) > 100
    pr = meta["present"]
    assert (pr["cellW"], pr["cellH"]) == (8, 16) and abs(pr["gamma"] - 1.2) < 1e-12
    atlas = np.fromfile(os.path.join(ROOT, "tests", "golden", "atlas_8x16.bin"), dtype=np.uint8)
    small = np.fromfile(tmp_path / "cube_small.rgba", dtype=np.uint8)
    img = np.fromfile(tmp_path / "cube_small.image", dtype=np.uint8).reshape(36 * 16, 96 * 8, 4)
    want = oracle.present(small, oracle.quantize(small, 96, 36, ramp, 1, 2, 12), 96, 36, atlas, 8, 16, 1.2, False, False)
    assert np.array_equal(img, want)
    ui = np.fromfile(tmp_path / "ico_ui.rgba", dtype=np.uint8)
    codes = np.fromfile(tmp_path / "ico_ui.codes", dtype=np.uint8).reshape(360, 640)
    assert np.array_equal(codes, oracle.quantize(ui, 640, 360, ramp, 1, 2, 12))
    colors = np.fromfile(tmp_path / "ico_ui.colors", dtype=np.uint32).reshape(360, 640)
    assert np.array_equal(colors, oracle.quantize_colors(ui, 640, 360, len(ramp)))
    assert np.unique(colors).size > 10 and not colors[5, 10:21].any()      # the UI text cells are black

    # quads + atlas through the JS backend (renderQuads): geometry and texel rows as the JS host built them
    import base64
    tx = meta["textured"]
    cols, rows = tx["cols"], tx["rows"]
    with open(os.path.join(ROOT, "tests", "golden", "ref_pathtrace_uv.json")) as fh:
        case = next(c for c in json.load(fh)["cases"] if c["name"] == "default_scene_atlas32")
    atlas = np.frombuffer(base64.b64decode(case["atlas_rgba"]), dtype=np.uint8).reshape(32, 32, 4)
    vbo = np.fromfile(tmp_path / "poster.vbo", dtype=np.float32).reshape(-1, 9)
    idx = np.fromfile(tmp_path / "poster.idx", dtype=np.uint32)
    uv8 = np.fromfile(tmp_path / "poster.uv8", dtype=np.uint16).reshape(-1, 8)
    py_vbo, py_idx, py_uv8 = pkg.host.build_geometry(case["scene"], textured=True)
    assert np.array_equal(idx, py_idx) and np.array_equal(uv8, py_uv8) and np.allclose(vbo, py_vbo, atol=1e-6)
    proj, view = pkg.host.camera_matrices(tx["cam"], cols, rows, tx["pixelAspect"])
    lights = pkg.host.pack_lights(case["scene"])
    rgba, depth, _ = oracle.render(vbo, idx, proj, view, lights, cols, rows, uv8=uv8, atlas=atlas)
    got = np.fromfile(tmp_path / "poster.rgba", dtype=np.uint8).reshape(rows, cols, 4)
    assert np.array_equal(got, rgba)
    assert np.array_equal(np.fromfile(tmp_path / "poster.depth", dtype=np.float32).view(np.uint32), depth.view(np.uint32).reshape(-1))
    assert np.array_equal(np.fromfile(tmp_path / "poster.ascii", dtype=np.uint8).reshape(rows, cols), oracle.quantize(rgba, cols, rows, ramp, 1, 2, 12))
    glyph = (rgba[..., 3] >= 32) & (rgba[..., 3] <= 126)
    assert glyph.sum() > 100
    ref_vbo, ref_idx = pkg.host.build_geometry(case["scene"])
    plain = oracle.render(ref_vbo, ref_idx, proj, view, lights, cols, rows)[0]
    got_plain = np.fromfile(tmp_path / "poster_plain.rgba", dtype=np.uint8).reshape(rows, cols, 4)
    # (sphere vertices come from libm sin/cos, which V8 and glibc may round differently in the last place)
    assert np.mean(got_plain == plain) > 0.999 and np.all(got_plain[..., 3] == 255)

    # a cube spun by model matrices built in JavaScript: the same matrices through the Python twin and the oracle
    spin = meta["spin"]
    vbo, idx = pkg.host.mesh_to_vbo(*pkg.geometry.cube(0.5))
    proj, view = pkg.host.camera_matrices(spin["cam"], spin["cols"], spin["rows"], spin["pixelAspect"])
    for k, M in enumerate(spin["frames"]):
        M = np.asarray(M, dtype=np.float32)
        want = pkg.geometry.spin_matrix(2 * np.pi * k / 6 + 0.1, (0.3, 1.0, 0.2), (0.1 * k, 0, 0), 1.0 + 0.05 * k)
        assert np.allclose(M, want, atol=1e-6)
        rgba, depth, _ = oracle.render(vbo, idx, proj, view, pkg.host.pack_lights({}), spin["cols"], spin["rows"], model=M)
        assert np.array_equal(np.fromfile(tmp_path / ("spin%d.rgba" % k), dtype=np.uint8).reshape(rgba.shape), rgba), k
        assert np.array_equal(np.fromfile(tmp_path / ("spin%d.depth" % k), dtype=np.float32).view(np.uint32), depth.view(np.uint32).reshape(-1)), k
        assert np.array_equal(np.fromfile(tmp_path / ("spin%d.ascii" % k), dtype=np.uint8).reshape(depth.shape),
                              oracle.quantize(rgba, spin["cols"], spin["rows"], ramp, 1, 2, 12)), k
        assert np.count_nonzero(depth < 1.0) > 100

    # frames in flight (renderAsync / collect): five cameras, three outstanding at a time, each against the oracle
    b = meta["burst"]
    cols, rows = b["cols"], b["rows"]
    (P, I), scene = meshes["ico"]
    vbo, idx = pkg.host.mesh_to_vbo(P, I)
    lights = pkg.host.pack_lights(scene)
    assert len(b["cams"]) == 5
    for k, cam in enumerate(b["cams"]):
        proj, view = pkg.host.camera_matrices(cam, cols, rows, b["pixelAspect"])
        rgba, depth, ids = oracle.render(vbo, idx, proj, view, lights, cols, rows)
        asc = oracle.quantize(rgba, cols, rows, ramp, 1, 2, 12)
        assert np.count_nonzero(depth < 1.0) > 1000
        assert np.array_equal(np.fromfile(tmp_path / ("burst%d.ascii" % k), dtype=np.uint8).reshape(rows, cols), asc), k
        # (the same camera again in the batch of five that shared one chain of launches)
        assert np.array_equal(np.fromfile(tmp_path / ("batch%d.rgba" % k), dtype=np.uint8).reshape(rows, cols, 4), rgba), k
        assert np.array_equal(np.fromfile(tmp_path / ("batch%d.depth" % k), dtype=np.float32).view(np.uint32).reshape(rows, cols), depth.view(np.uint32)), k
        assert np.array_equal(np.fromfile(tmp_path / ("batch%d.ascii" % k), dtype=np.uint8).reshape(rows, cols), asc), k
        if k % 3 != 2:
            assert np.array_equal(np.fromfile(tmp_path / ("burst%d.rgba" % k), dtype=np.uint8).reshape(rows, cols, 4), rgba), k
            assert np.array_equal(np.fromfile(tmp_path / ("burst%d.depth" % k), dtype=np.float32).view(np.uint32).reshape(rows, cols),
                                  depth.view(np.uint32)), k
    assert {"k_setup", "k_fill", "k_tile", "k_ascii"} <= set(b["kernels"]), b["kernels"]
    assert all(b["kernels"][k]["launches"] == 5 and b["kernels"][k]["ms"] > 0 for k in ("k_setup", "k_fill", "k_tile", "k_ascii")), b["kernels"]
    # last camera again: ids, codes and colours read on demand; then rows 100..199 only
    assert np.array_equal(np.fromfile(tmp_path / "burst_last.ids", dtype=np.uint32).reshape(rows, cols), ids)
    assert np.array_equal(np.fromfile(tmp_path / "burst_last.ascii", dtype=np.uint8).reshape(rows, cols), asc)
    assert np.array_equal(np.fromfile(tmp_path / "burst_last.colors", dtype=np.uint32).reshape(rows, cols),
                          oracle.quantize_colors(rgba, cols, rows, len(ramp)))
    strip = np.fromfile(tmp_path / "burst_strip.ascii", dtype=np.uint8).reshape(rows, cols)
    assert np.array_equal(strip[100:200], asc[100:200])

    # the group of devices driven from Node: the first three burst cameras again, gathered in batches of two
    for k in range(3):
        proj, view = pkg.host.camera_matrices(b["cams"][k], cols, rows, b["pixelAspect"])
        rgba_k, _d, _i = oracle.render(vbo, idx, proj, view, lights, cols, rows)
        assert np.array_equal(np.fromfile(tmp_path / ("group%d.ascii" % k), dtype=np.uint8).reshape(rows, cols),
                              oracle.quantize(rgba_k, cols, rows, ramp, 1, 2, 12)), k

    # the ray backend through the router: same image as the oracle fed with the reference's recorded uniforms
    rtm = meta["raytrace"]
    with open(os.path.join(ROOT, "tests", "golden", "ref_raytrace_uniforms.json")) as fh:
        rcase = next(c for c in json.load(fh)["cases"] if c["scene"] == rtm["scene"] and c["cam_name"] == rtm["cam_name"])
    u, cam = rt_uniforms_of(rcase)
    cols, rows = rtm["cols"], rtm["rows"]
    want = oracle.rt_render(u, cols, rows, cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
    got = np.fromfile(tmp_path / "rt.rgba", dtype=np.uint8).reshape(rows, cols, 4)
    assert np.array_equal(got, want) and len(np.unique(want.reshape(-1, 4), axis=0)) > 50
    assert np.array_equal(np.fromfile(tmp_path / "rt.ascii", dtype=np.uint8).reshape(rows, cols), oracle.quantize(want, cols, rows, ramp, 1, 2, 12))
