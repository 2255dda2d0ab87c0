"""Mesh generators for tests and benchmarks (Python twin of ascii-renderer_amd/js/geometry.js).

The reference ships an empty js/geometry.js, so these are this project's own generators.  They emit
the interchange shape SceneBuilder.addMesh accepts (js/render/scene_api.js:169-192): `positions`
(3 floats per vertex) and `indices` (3 per triangle, counter-clockwise seen from outside).
Both language twins draw from the same mulberry32 stream in the same order and use the same
literal constants, so they produce bit-identical binary32 meshes.
"""
import math

import numpy as np

TAN40 = 0.8390996311772799       # tan(40 deg): half of the reference's fixed 80 deg vertical fov
_GOLD = 0x6D2B79F5


def mulberry32_stream(seed, n, offset=0):
    """n consecutive outputs of mulberry32(seed) in [0,1) as float64, starting at draw `offset`."""
    k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
    a = (np.uint64(seed & 0xFFFFFFFF) + k * np.uint64(_GOLD)) & np.uint64(0xFFFFFFFF)
    a = a.astype(np.uint32)
    with np.errstate(over="ignore"):
        t = (a ^ (a >> np.uint32(15))) * (np.uint32(1) | a)
        t = (t + (t ^ (t >> np.uint32(7))) * (np.uint32(61) | t)) ^ t
        r = t ^ (t >> np.uint32(14))
    return r.astype(np.float64) / 4294967296.0


def cube(half=0.5):
    """Axis-aligned cube centred at the origin: 8 vertices, 12 outward-CCW triangles."""
    h = float(half)
    P = np.array([[-h, -h, -h], [h, -h, -h], [h, h, -h], [-h, h, -h],
                  [-h, -h, h], [h, -h, h], [h, h, h], [-h, h, h]], dtype=np.float32)
    I = np.array([[4, 5, 6], [4, 6, 7],      # +z
                  [1, 0, 3], [1, 3, 2],      # -z
                  [5, 1, 2], [5, 2, 6],      # +x
                  [0, 4, 7], [0, 7, 3],      # -x
                  [7, 6, 2], [7, 2, 3],      # +y
                  [0, 1, 5], [0, 5, 4]],     # -y
                 dtype=np.uint32)
    return P, I


def icosphere(subdiv=4, radius=1.0):
    """Unit icosahedron subdivided `subdiv` times (20 * 4**subdiv triangles), outward CCW."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
             (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    s = math.sqrt(1.0 + t * t)
    verts = [(v[0] / s, v[1] / s, v[2] / s) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(int(subdiv)):
        cache, out = {}, []

        def mid(i, j):
            key = (i, j) if i < j else (j, i)
            if key not in cache:
                a, b = verts[key[0]], verts[key[1]]
                m = ((a[0] + b[0]) * 0.5, (a[1] + b[1]) * 0.5, (a[2] + b[2]) * 0.5)
                L = math.sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2])
                verts.append((m[0] / L, m[1] / L, m[2] / L))
                cache[key] = len(verts) - 1
            return cache[key]

        for (a, b, c) in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            out += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = out
    P = (np.asarray(verts, dtype=np.float64) * float(radius)).astype(np.float32)
    return P, np.asarray(faces, dtype=np.uint32)


def spin_matrix(angle, axis=(0.0, 1.0, 0.0), translate=(0.0, 0.0, 0.0), scale=1.0):
    """Column-major model matrix (float32[16]): uniform scale, then rotation by `angle` about `axis`
    (Rodrigues' formula in binary64), then translation - what a spinning mesh hands to ar_set_model per frame."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / math.sqrt(float(ax @ ax))
    x, y, z = ax
    c, s = math.cos(angle), math.sin(angle)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]]) * float(scale)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = translate
    return M.T.reshape(-1).astype(np.float32)


def random_mesh(seed, n_tris, cols, rows, pixel_aspect=1.0, depth_range=(2.0, 60.0), cells=2.0,
                chunk=1 << 20):
    """Screen-uniform cloud of small random triangles in front of a camera at the origin looking
    along +X (yaw = pitch = 0; screen right = +Z, up = +Y).

    Per triangle, 15 consecutive mulberry32 draws: depth x in depth_range; centre y, z uniform over
    the frustum cross-section at that depth; three vertices = centre + x*rho*u with u uniform in
    [-1,1]^3 and rho = cells * (2*TAN40/rows) (each vertex within `cells` cells of the centre);
    then an albedo in [0.2, 1]^3.  Winding is as generated (about half the triangles are back
    faces).  Returns (positions float32[T*3,3], indices uint32[T,3], colors float32[T,3]).
    """
    aspect = (cols / rows) * pixel_aspect
    rho = cells * (2.0 * TAN40 / rows)
    d0, d1 = depth_range
    P = np.empty((n_tris * 3, 3), dtype=np.float32)
    C = np.empty((n_tris, 3), dtype=np.float32)
    for s in range(0, n_tris, chunk):
        e = min(n_tris, s + chunk)
        u = mulberry32_stream(seed, (e - s) * 15, offset=s * 15).reshape(-1, 15)
        x = d0 + (d1 - d0) * u[:, 0]
        cy = (2.0 * u[:, 1] - 1.0) * x * TAN40
        cz = (2.0 * u[:, 2] - 1.0) * x * TAN40 * aspect
        centre = np.stack([x, cy, cz], axis=1)
        off = (2.0 * u[:, 3:12] - 1.0).reshape(-1, 3, 3) * (x * rho)[:, None, None]
        P[s * 3:e * 3] = (centre[:, None, :] + off).reshape(-1, 3)
        C[s:e] = 0.2 + 0.8 * u[:, 12:15]
    I = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    return P, I, C
