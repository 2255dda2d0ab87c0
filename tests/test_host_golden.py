"""Python host mirror vs values recorded from the reference's own host code
(tests/golden/ref_raster_host.json, produced by tests/golden/gen_from_reference.mjs)."""
import base64
import json
import os

import numpy as np
import pytest


def _f32(b):
    return np.frombuffer(base64.b64decode(b), dtype=np.float32)


@pytest.fixture(scope="module")
def ref(golden_dir):
    with open(os.path.join(golden_dir, "ref_raster_host.json")) as fh:
        return json.load(fh)


def test_geometry_matches_reference_vbo_ibo(pkg, ref):
    for case in ref["cases"]:
        scene = ref["scenes"][case["scene"]]
        vbo, idx = pkg.host.build_geometry(scene)
        want_v = _f32(case.get("vbo_f32", ""))
        want_i = np.frombuffer(base64.b64decode(case.get("ibo_u32", "")), dtype=np.uint32)
        assert idx.size == want_i.size, case["scene"]
        assert np.array_equal(idx, want_i), case["scene"]
        assert vbo.size == want_v.size, case["scene"]
        got = vbo.reshape(-1)
        # binary64 libm results (sin/cos/hypot) may differ by an ulp between V8 and glibc before the
        # binary32 store; allow one binary32 ulp and require (nearly) all values identical
        diff = got.view(np.int32).astype(np.int64) - want_v.view(np.int32).astype(np.int64)
        near_zero = (np.abs(got) < 1e-15) & (np.abs(want_v) < 1e-15)
        assert np.all((np.abs(diff) <= 1) | near_zero), case["scene"]
        assert np.mean(diff == 0) > 0.999 or got.size < 2000, case["scene"]


def test_default_scene_shape(pkg, ref):
    vbo, idx = pkg.host.build_geometry(ref["scenes"]["default_scene"])
    assert vbo.shape == (442, 9) and idx.size == 2304          # two spheres; quads are ignored
    assert np.allclose(vbo[:, 6:9], 0.8)                        # builder scenes fall back to grey


def test_camera_matrices_match_reference(pkg, ref):
    for case in ref["cases"]:
        if "uProj" not in case["uniforms"]:
            continue
        proj, view = pkg.host.camera_matrices(case["camera"], case["cols"], case["rows"], case["pixelAspect"])
        assert np.allclose(proj, _f32(case["uniforms"]["uProj"]), rtol=0, atol=0) or \
            np.max(np.abs(proj.view(np.int32) - _f32(case["uniforms"]["uProj"]).view(np.int32))) <= 1
        want = _f32(case["uniforms"]["uView"])
        assert np.allclose(view, want, rtol=2e-7, atol=1e-7), (case["scene"], case["cam_name"])


def test_known_projection_values(pkg):
    proj, view = pkg.host.camera_matrices({"pos": (0, 1.5, 6), "yaw": 0.0, "pitch": 0.0}, 96, 36, None)
    assert abs(proj[0] - 0.446908) < 1e-6 and abs(proj[5] - 1.191754) < 1e-6
    assert abs(proj[10] + 1.001001) < 1e-6 and proj[11] == -1 and abs(proj[14] + 0.10005) < 1e-6
    assert np.allclose(view, [0, 0, -1, 0, 0, 1, 0, 0, 1, 0, 0, 0, -6, -1.5, 0, 1], atol=1e-7)


def test_light_uniforms_match_reference(pkg, ref):
    for case in ref["cases"]:
        u = case["uniforms"]
        if "uAmbient" not in u:
            continue
        L = pkg.host.pack_lights(ref["scenes"][case["scene"]])
        assert np.array_equal(L["ambient"], _f32(u["uAmbient"])), case["scene"]
        assert np.array_equal(L["dir"], _f32(u["uLightDir"])), case["scene"]
        assert np.array_equal(L["dir_color"], _f32(u["uLightColor"])), case["scene"]
        n = u["uPLCount"]
        assert L["pl_pos"].shape[0] == n
        assert np.array_equal(L["pl_pos"].reshape(-1), _f32(u["uPLPos[0]"])[: 3 * n])
        assert np.array_equal(L["pl_col"].reshape(-1), _f32(u["uPLCol[0]"])[: 3 * n])
        assert not np.any(_f32(u["uPLPos[0]"])[3 * n:]) and not np.any(_f32(u["uPLCol[0]"])[3 * n:])


def test_gl_state_recorded_from_reference(ref):
    c = ref["cases"][0]
    assert c["enabled"] == ["DEPTH_TEST", "CULL_FACE"] and c["cullFace"] == "BACK"
    assert c["clearColor"] == [0, 0, 0, 1] and c["depthFormat"] == "DEPTH_COMPONENT16"
    assert c["draw"] == {"mode": "TRIANGLES", "count": 2304, "type": "UNSIGNED_INT", "offset": 0}
