#!/usr/bin/env python3
"""Writes tests/golden/self_consistency_frames.npz: frames of the CPU oracle (oracle/cpu_raster.c) frozen as
data - C1 (the cube, 80x40, sixty orbit angles), C2 (the icosphere, 640x360, without and with point lights), a soup
of triangles through the near plane (200x100), the cube under a model matrix, the reference's default scene (two
smooth spheres, 96x36), that scene textured (quads + atlas, 120x50) and 64 k small triangles shaped like C3 (320x180):
ASCII codes, binary32 depth and RGBA8 per frame.  Frames already in the file must come out unchanged (the script
refuses to overwrite one that differs unless given --force).

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
    from test_oracle_second_opinion import GOLDEN, _oracle_frame, golden_cases
    orc.build()
    out = {}
    for name, sc in golden_cases(ar):
        out[name + "_rgba"], out[name + "_depth"], out[name + "_ascii"] = _oracle_frame(ar, orc, sc)
    if os.path.exists(GOLDEN) and "--force" not in sys.argv:
        old = np.load(GOLDEN)
        changed = [k for k in old.files if k in out and not np.array_equal(old[k].view(np.uint8), np.ascontiguousarray(out[k]).view(np.uint8))]
        if changed:
            raise SystemExit("frozen frames would change: %s" % ", ".join(changed[:8]))
    np.savez_compressed(GOLDEN, **out)
    print("wrote %s: %d arrays, %d bytes" % (GOLDEN, len(out), os.path.getsize(GOLDEN)))


if __name__ == "__main__":
    main()
