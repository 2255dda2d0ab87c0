"""Ray backend: HIP kernel (through the C ABI) against the CPU oracle, bit for bit, on the uniform arrays
recorded from the reference's RaytraceBackend (tests/golden/ref_raytrace_uniforms.json) and on larger
synthetic scenes.  RGBA8 and ASCII codes must be identical."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FOVY = 80 * math.pi / 180
RAMP = "@%#*+=-:. "


@pytest.fixture(scope="module")
def R(pkg):
    r = pkg.Renderer(0)
    yield r
    r.close()


def _check(oracle, R, u, cols, rows, cam, mode=(True, 2, 12), label=""):
    R.set_ascii(RAMP, *mode)
    R.rt_set_scene(u)
    out = R.rt_render(cols, rows, cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
    want = oracle.rt_render(u, cols, rows, cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
    assert np.array_equal(out["rgba"], want), "%s: %d bytes differ" % (label, np.count_nonzero(out["rgba"] != want))
    assert np.array_equal(out["ascii"], oracle.quantize(want, cols, rows, RAMP, int(mode[0]), mode[1], mode[2])), label
    return out


def test_reference_uniform_scenes(oracle, R, golden_dir, rt_uniforms_of):
    with open(os.path.join(golden_dir, "ref_raytrace_uniforms.json")) as fh:
        ref = json.load(fh)
    varied = 0
    for case in ref["cases"]:
        u, cam = rt_uniforms_of(case)
        label = case["scene"] + "/" + case["cam_name"]
        out = _check(oracle, R, u, case["cols"], case["rows"], cam, label=label)
        # the same scene on a finer grid and without smoothing
        _check(oracle, R, u, 333, 77, cam, mode=(False, 2, 12), label=label + " 333x77")
        varied += len(np.unique(out["rgba"].reshape(-1, 4), axis=0)) > 8
    assert varied >= 5


def _synthetic(seed, n_s, n_t, n_mats=12):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-6, 6, size=(n_s, 3)); c[:, 1] = np.abs(c[:, 1]) * 0.5 + 0.3
    spheres = np.concatenate([c, rng.uniform(0.2, 0.9, size=(n_s, 1))], axis=1)
    tc = rng.uniform(-7, 7, size=(n_t, 1, 3)); tc[:, :, 1] = np.abs(tc[:, :, 1]) * 0.6
    tri = tc + rng.uniform(-1.2, 1.2, size=(n_t, 3, 3))
    albedo = rng.uniform(0.1, 1.0, size=(n_mats, 3))
    refl = (rng.uniform(size=n_mats) < 0.3).astype(np.int32)
    return {"uNumS": n_s, "uSpheres": spheres, "uSm": rng.integers(-1, n_mats + 2, size=n_s),
            "uNumP": 3, "uPlanes": [0, 1, 0, 0, 0, 0, 1, 9, 1, 0, 0, 9], "uPm": [0, 1, 2],
            "uNumT": n_t, "uTa": tri[:, 0], "uTb": tri[:, 1], "uTc": tri[:, 2], "uTm": rng.integers(0, n_mats, size=n_t),
            "uNumMats": n_mats, "uMatAlbedo": albedo, "uMatReflective": refl,
            "uNumPL": 3, "uPLPos": [3, 6, 2, -4, 5, -3, 0, 8, 0], "uPLCol": [20, 18, 15, 6, 8, 14, 10, 10, 10],
            "uNumDL": 2, "uDLDir": [0.3, -1, 0.2, -0.5, -0.7, -0.4], "uDLCol": [0.5, 0.5, 0.4, 0.2, 0.2, 0.3], "uEnv": (0.25, 0.35, 0.6)}


def test_synthetic_scenes_at_full_size(oracle, R):
    """The reference's limits (32 spheres, 64 triangles) and well beyond them, 1920x1080 and odd grids."""
    u = _synthetic(5, 32, 64)
    cam = {"pos": (7.5, 3.0, 6.5), "yaw": -2.4, "pitch": -0.3, "fovy": FOVY, "pix_aspect": 1.0}
    out = _check(oracle, R, u, 1920, 1080, cam, label="32s/64t 1080p")
    assert len(np.unique(out["rgba"].reshape(-1, 4), axis=0)) > 500
    big = _synthetic(6, 200, 1500, 40)
    for cols, rows, mode in ((257, 130, (True, 3, 20)), (640, 360, (True, 1, 4)), (1, 1, (True, 2, 12))):
        _check(oracle, R, big, cols, rows, dict(cam, pix_aspect=0.5), mode=mode, label="200s/1500t %dx%d" % (cols, rows))


def test_strips_async_and_interleaving_with_raster_frames(pkg, oracle, R):
    u = _synthetic(7, 20, 40)
    cam = {"pos": (6.0, 2.5, -5.0), "yaw": 2.3, "pitch": -0.25, "fovy": FOVY, "pix_aspect": 0.5}
    cols, rows = 200, 96
    full = _check(oracle, R, u, cols, rows, cam)
    R.set_strip(24, 72)
    part = R.rt_render(cols, rows, cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
    R.set_strip(0, 0)
    for k in ("rgba", "ascii"):
        assert np.array_equal(part[k][24:72], full[k][24:72]), k
    # raster frames in flight, then a ray frame, then raster again: each reads back what it drew
    sc = pkg.scenes.icosphere_scene(angle=0.4, cols=cols, rows=rows, subdiv=3)
    pkg.scenes.upload(R, sc, RAMP, True, 2, 12)
    for _ in range(4):
        R.render_async(cols, rows)
    R.rt_render_async(cols, rows, cam["pos"], cam["yaw"], cam["pitch"], cam["fovy"], cam["pix_aspect"])
    R.sync()
    got = R.read()
    assert np.array_equal(got["rgba"], full["rgba"]) and np.array_equal(got["ascii"], full["ascii"])
    ras = R.render(cols, rows)
    want = oracle.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], cols, rows)[0]
    assert np.array_equal(ras["rgba"], want)


def test_no_scene_and_bad_arguments(pkg):
    r = pkg.Renderer(0)
    rgba = np.full((4, 4, 4), 9, dtype=np.uint8)
    r._check(r._L.ar_rt_render(r._h, 4, 4, np.zeros(3, dtype=np.float32).ctypes.data_as(pkg.binding._fp), 0.0, 0.0, FOVY, 1.0,
                               rgba.ctypes.data_as(pkg.binding._u8p), None))
    assert np.all(rgba == 9)                                       # raytrace.js:321: no scene, nothing drawn
    with pytest.raises(pkg.AsciiRastError):
        r.rt_render(0, 5, (0, 0, 0), 0.0, 0.0, FOVY)
    with pytest.raises(pkg.AsciiRastError):
        r.rt_set_scene({"uNumS": 2})                               # count without arrays
    r.close()
