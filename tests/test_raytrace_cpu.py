"""Ray backend on the CPU side: the oracle's restatement of the reference's single-bounce ray tracer
(raytrace_shader.js) checked on scenes whose images follow from the shader text by hand.  The reference
holds no output of its GLSL, so this stage is PARITY UNPINNED and the oracle is its definition; the uniform
arrays it consumes are pinned by tests/golden/ref_raytrace_uniforms.json (see tests/test_js_host.py)."""
import base64
import json
import math
import os

import numpy as np
import pytest

FOVY = 80 * math.pi / 180


def scene(**kw):
    u = {"uNumS": 0, "uNumP": 0, "uNumT": 0, "uNumMats": 1, "uMatAlbedo": [0.8, 0.8, 0.8], "uMatReflective": [0],
         "uNumPL": 0, "uNumDL": 0, "uEnv": (0, 0, 0)}
    u.update(kw)
    return u


def test_camera_basis(oracle):
    uu, vv, ww, focal = oracle.rt_camera(0.0, 0.0, FOVY)
    assert ww.tolist() == [1, 0, 0] and uu.tolist() == [0, 0, 1] and vv.tolist() == [0, 1, 0]   # yaw 0 looks along +X
    assert abs(focal - 1 / math.tan(FOVY / 2)) < 1e-6
    uu, vv, ww, _ = oracle.rt_camera(-math.pi / 2, 0.0, FOVY)
    assert np.allclose(ww, [0, 0, -1], atol=1e-6) and np.allclose(uu, [1, 0, 0], atol=1e-6)
    uu, vv, ww, _ = oracle.rt_camera(0.3, math.pi / 2, FOVY)     # straight up: cross(ww, up) all but vanishes, still a unit vector
    assert abs(np.linalg.norm(uu) - 1) < 1e-5 and abs(np.linalg.norm(vv) - 1) < 1e-5


def test_miss_shows_the_environment(oracle):
    rgba = oracle.rt_render(scene(uEnv=(0.2, 0.5, 2.0)), 16, 9, (0, 0, 0), 0.0, 0.0, FOVY)
    assert np.all(rgba == np.array([51, 128, 255, 255], dtype=np.uint8))


def test_wall_lit_head_on_is_uniform(oracle):
    """Plane x = 5 with normal (-1, 0, 0), directional light travelling along +X: N.L = 1 in every cell, so the
    whole image is albedo * light colour; a negative material index turns the wall into a miss."""
    u = scene(uNumP=1, uPlanes=[-1, 0, 0, 5], uPm=[1], uNumMats=2, uMatAlbedo=[0.8, 0.8, 0.8, 0.5, 0.25, 1.0], uMatReflective=[0, 0],
              uNumDL=1, uDLDir=[2, 0, 0], uDLCol=[0.8, 0.8, 0.5], uEnv=(1, 0, 0))
    rgba = oracle.rt_render(u, 40, 20, (0, 0, 0), 0.0, 0.0, FOVY, 0.5)
    assert np.all(rgba == np.array([102, 51, 128, 255], dtype=np.uint8))      # 0.4, 0.2, 0.5
    u["uPm"] = [7]                                                             # beyond the table: clamps to its last entry
    assert np.all(oracle.rt_render(u, 40, 20, (0, 0, 0), 0.0, 0.0, FOVY, 0.5) == np.array([102, 51, 128, 255], dtype=np.uint8))
    u["uPm"] = [-1]
    assert np.all(oracle.rt_render(u, 40, 20, (0, 0, 0), 0.0, 0.0, FOVY, 0.5) == np.array([255, 0, 0, 255], dtype=np.uint8))
    u["uPm"], u["uNumMats"], u["uMatAlbedo"], u["uMatReflective"] = [0], 0, [], []    # empty table: albedo 0.8
    assert np.all(oracle.rt_render(u, 40, 20, (0, 0, 0), 0.0, 0.0, FOVY, 0.5) == np.array([163, 163, 102, 255], dtype=np.uint8))


def test_sphere_is_mirror_symmetric_and_casts_a_shadow(oracle):
    """Sphere on the view axis, point light on the axis above it, floor of two triangles: the image is its own
    left-right mirror image; the floor below the sphere is black (shadow), further out it is lit; a plane in
    the light's way casts no shadow (raytrace_shader.js:159 leaves planes out of the shadow test)."""
    floor = dict(uNumT=2, uTa=[-50, -1, -50, -50, -1, -50], uTb=[50, -1, -50, 50, -1, 50], uTc=[50, -1, 50, -50, -1, 50], uTm=[0, 0])
    u = scene(uNumS=1, uSpheres=[6, 0, 0, 1], uSm=[0], uNumPL=1, uPLPos=[6, 6, 0], uPLCol=[30, 30, 30], **floor)
    cols, rows = 64, 32
    rgba = oracle.rt_render(u, cols, rows, (0, 0.5, 0), 0.0, -0.15, FOVY, 0.5)
    assert np.array_equal(rgba, rgba[:, ::-1])
    lum = rgba[..., :3].astype(int).sum(axis=2)
    assert lum.max() > 300
    # trace the shadow by hand: floor point (6, -1, 0) is straight below the sphere -> black
    ys, xs = np.nonzero(lum == 0)
    assert ys.size > 10 and abs(xs.mean() - (cols - 1) / 2) < 1.0
    u2 = dict(u, uNumP=1, uPlanes=[0, -1, 0, 4], uPm=[0])       # ceiling y = 4 between light and scene, facing down
    below = oracle.rt_render(u2, cols, rows, (0, 0.5, 0), 0.0, -0.15, FOVY, 0.5)
    assert np.array_equal(below[rows // 2:], rgba[rows // 2:])  # lower half (no ceiling in view) is unchanged


def test_mirror_sphere_alone_reflects_the_environment(oracle):
    u = scene(uNumS=1, uSpheres=[5, 0, 0, 2], uSm=[0], uMatAlbedo=[1, 1, 1], uMatReflective=[1], uEnv=(0.1, 0.6, 0.9),
              uNumPL=1, uPLPos=[0, 5, 0], uPLCol=[9, 9, 9])
    rgba = oracle.rt_render(u, 32, 16, (0, 0, 0), 0.0, 0.0, FOVY, 0.5)
    assert np.all(rgba == np.array([26, 153, 230, 255], dtype=np.uint8))
    # a diffuse wall behind the camera shows up in the mirror
    u2 = dict(u, uNumP=1, uPlanes=[1, 0, 0, 3], uPm=[1], uNumMats=2, uMatAlbedo=[1, 1, 1, 1, 0, 0], uMatReflective=[1, 0])
    seen = oracle.rt_render(u2, 32, 16, (0, 0, 0), 0.0, 0.0, FOVY, 0.5)
    centre = seen[8, 16]
    assert centre[0] > 0 and centre[1] == 0 and centre[2] == 0


def test_golden_uniform_scenes_render(oracle, golden_dir, rt_uniforms_of):
    """The scenes of the fixture produce non-trivial images (they are the GPU parity inputs)."""
    with open(os.path.join(golden_dir, "ref_raytrace_uniforms.json")) as fh:
        ref = json.load(fh)
    kinds = set()
    for case in ref["cases"]:
        u, cam = rt_uniforms_of(case)
        rgba = oracle.rt_render(u, case["cols"], case["rows"], cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
        kinds.add((case["scene"], len(np.unique(rgba.reshape(-1, 4), axis=0)) > 8))
    assert ("unified", True) in kinds and ("legacy", True) in kinds and ("crowded", True) in kinds
