"""JavaScript host (router, HIP backend host half, scene builder, geometry twin) checked under Node
against the values recorded from the reference, and against the Python twins.  No GPU involved."""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def js_results(tmp_path_factory, golden_dir):
    if NODE is None:
        pytest.skip("node is not installed")
    out = str(tmp_path_factory.mktemp("js") / "results.json")
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_host.mjs"), golden_dir, out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    with open(out) as fh:
        return json.load(fh)


def test_js_host_matches_reference_recordings(js_results):
    assert js_results["checks"] > 60


def test_js_and_python_generators_are_bit_identical(js_results, pkg):
    h = js_results["hashes"]
    P, I = pkg.geometry.cube(0.5)
    assert h["cube_pos"] == _sha(P) and h["cube_idx"] == _sha(I)
    P, I = pkg.geometry.icosphere(3, 1.0)
    assert h["ico_pos"] == _sha(P) and h["ico_idx"] == _sha(I)
    assert h["ico_vbo"] == _sha(pkg.host.mesh_to_vbo(P, I)[0])
    P, I, C = pkg.geometry.random_mesh(0xA5C11, 5000, 1920, 1080, 1.0)
    assert h["rm_pos"] == _sha(P) and h["rm_col"] == _sha(C)
    assert h["rm_vbo"] == _sha(pkg.host.mesh_to_vbo(P, I, C)[0])
    assert js_results["mulberry"] == list(pkg.geometry.mulberry32_stream(12345, 3))


def test_js_and_python_texel_coordinate_rows_are_identical(js_results, pkg, golden_dir):
    with open(os.path.join(golden_dir, "ref_pathtrace_uv.json")) as fh:
        cases = json.load(fh)["cases"]
    for c in cases:
        vbo, idx, uv8 = pkg.host.build_geometry(c["scene"], textured=True)
        got = js_results["textured"][c["name"]]
        assert got["uv8"] == _sha(uv8) and got["idx"] == _sha(idx) and got["verts"] == vbo.shape[0], c["name"]
