#!/usr/bin/env python3
"""Writes tests/golden/self_consistency_frames.npz: frames of the CPU oracle (oracle/cpu_raster.c) frozen as
data - C1 (the cube, 80x40, sixty orbit angles) and C2 (the icosphere, 640x360, without and with point lights):
ASCII codes, binary32 depth and RGBA8 per frame.

These are SELF-CONSISTENCY fixtures: the reference runs its raster stage inside WebGL and holds no expected output
for it, so the frames cannot come from the reference.  They pin the oracle against drift (a change that alters the
oracle and the kernels together no longer passes silently) and give the HIP path a target with no oracle in the
loop.  The oracle itself is cross-checked by tests/test_oracle_second_opinion.py.

  python tests/golden/gen_self_consistency.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ar = importlib.import_module("ascii-renderer_amd")
    from oracle import oracle as orc
    from test_oracle_second_opinion import GOLDEN, golden_cases
    orc.build()
    out = {}
    for name, sc in golden_cases(ar):
        rgba, depth, _ids = orc.render(sc["vbo"], sc["idx"], sc["proj"], sc["view"], sc["lights"], sc["cols"], sc["rows"], want_ids=False)
        out[name + "_ascii"] = orc.quantize(rgba, sc["cols"], sc["rows"], ar.scenes.DEFAULT_RAMP, 1, 2, 12)
        out[name + "_depth"] = depth
        out[name + "_rgba"] = rgba
    np.savez_compressed(GOLDEN, **out)
    print("wrote %s: %d arrays, %d bytes" % (GOLDEN, len(out), os.path.getsize(GOLDEN)))


if __name__ == "__main__":
    main()
