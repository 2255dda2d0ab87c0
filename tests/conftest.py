import base64
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module("ascii-renderer_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle: the checker, never the thing under test in the parity tests."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _f32(b):
    import numpy as np
    return np.frombuffer(base64.b64decode(b), dtype=np.float32)


def _i32(b):
    import numpy as np
    return np.frombuffer(base64.b64decode(b), dtype=np.int32)


def _rt_uniforms_of(case):
    """One case of tests/golden/ref_raytrace_uniforms.json -> (dict keyed by uniform name with the arrays cut to
    their counts, camera dict)."""
    import numpy as np
    U = case["uniforms"]
    n = {k: int(U.get(k, 0)) for k in ("uNumS", "uNumP", "uNumT", "uNumMats", "uNumPL", "uNumDL")}
    u = dict(n)

    def take(key, count, width, dec):
        name = key + "[0]"
        u[key] = dec(U[name])[: count * width].copy() if name in U and count else np.zeros(0, dtype=np.float32)

    take("uSpheres", n["uNumS"], 4, _f32); take("uSm", n["uNumS"], 1, _i32)
    take("uPlanes", n["uNumP"], 4, _f32); take("uPm", n["uNumP"], 1, _i32)
    for k in ("uTa", "uTb", "uTc"):
        take(k, n["uNumT"], 3, _f32)
    take("uTm", n["uNumT"], 1, _i32)
    take("uMatAlbedo", n["uNumMats"], 3, _f32); take("uMatReflective", n["uNumMats"], 1, _i32)
    take("uPLPos", n["uNumPL"], 3, _f32); take("uPLCol", n["uNumPL"], 3, _f32)
    take("uDLDir", n["uNumDL"], 3, _f32); take("uDLCol", n["uNumDL"], 3, _f32)
    u["uEnv"] = _f32(U["uEnv"])
    cam = {"pos": _f32(U["uCamPos"]), "yaw": float(_f32(U["uYaw"])[0]), "pitch": float(_f32(U["uPitch"])[0]),
           "fovy": float(_f32(U["uFovY"])[0]), "pix_aspect": float(_f32(U["uPixAspect"])[0])}
    return u, cam


@pytest.fixture(scope="session")
def rt_uniforms_of():
    return _rt_uniforms_of
