"""The benchmark / parity workloads (configs C1..C5) as ready-to-upload buffers.

Every builder returns a dict with vbo (float32[V,9]), idx (uint32[T*3]), proj, view (float32[16]),
lights (see host.pack_lights), cols, rows and a short name.  Cameras and lights go through the same
host code paths as a scene handed to the backend (host.camera_matrices / host.pack_lights).
"""
import math

import numpy as np

from . import geometry, host

DEFAULT_RAMP = "@%#*+=-:. "          # js/config.js:15 (the later duplicate key wins)


def look_at_origin(eye):
    """yaw / pitch (raster.js:41-45 convention: yaw = pitch = 0 looks along +X) that face the origin."""
    ex, ey, ez = eye
    n = math.sqrt(ex * ex + ey * ey + ez * ez)
    return math.atan2(-ez, -ex), math.asin(-ey / n)


def orbit_camera(angle, radius=3.0, height=1.0):
    eye = (radius * math.cos(angle), height, radius * math.sin(angle))
    yaw, pitch = look_at_origin(eye)
    return {"pos": eye, "yaw": yaw, "pitch": pitch}


def _pack(name, vbo, idx, cam, cols, rows, pixel_aspect, scene=None):
    proj, view = host.camera_matrices(cam, cols, rows, pixel_aspect)
    return {"name": name, "vbo": vbo, "idx": idx, "proj": proj, "view": view,
            "lights": host.pack_lights(scene or {}), "cols": cols, "rows": rows, "camera": cam}


def cube_scene(angle=0.6, cols=80, rows=40, pixel_aspect=0.5):
    """C1: unit cube, camera orbiting at radius 3, height 1, default lights."""
    P, I = geometry.cube(0.5)
    vbo, idx = host.mesh_to_vbo(P, I)
    return _pack("cube", vbo, idx, orbit_camera(angle), cols, rows, pixel_aspect)


def icosphere_scene(angle=0.6, cols=640, rows=360, pixel_aspect=0.5, subdiv=4, point_lights=0):
    """C2: icosphere (20 * 4**subdiv triangles), same orbit camera."""
    P, I = geometry.icosphere(subdiv, 1.0)
    vbo, idx = host.mesh_to_vbo(P, I)
    scene = {}
    if point_lights:
        scene = {"lights": {"points": [
            {"p": [2.0 * math.cos(2.1 * k), 1.5, 2.0 * math.sin(2.1 * k)], "color": [1.0, 0.6 + 0.1 * k, 0.5], "intensity": 3.0}
            for k in range(point_lights)]}}
    return _pack("icosphere%d" % subdiv, vbo, idx, orbit_camera(angle), cols, rows, pixel_aspect, scene)


def random_scene(n_tris=1_000_000, cols=1920, rows=1080, seed=0xA5C11, point_lights=False, mesh_grid=(1920, 1080)):
    """C3 / C4 / C5: screen-uniform cloud of small triangles seen from the origin along +X.

    The mesh is generated for `mesh_grid` (so C4 renders the same scene as C3 on a finer grid).
    point_lights=True adds the four lights of C5 at (10, +-5, +-5), intensity 20, white.
    """
    P, I, C = geometry.random_mesh(seed, n_tris, mesh_grid[0], mesh_grid[1], 1.0)
    vbo, idx = host.mesh_to_vbo(P, I, C)
    scene = {}
    if point_lights:
        scene = {"lights": {"points": [{"p": [10.0, sy * 5.0, sz * 5.0], "color": [1, 1, 1], "intensity": 20.0}
                                       for sy in (1, -1) for sz in (1, -1)]}}
    cam = {"pos": (0.0, 0.0, 0.0), "yaw": 0.0, "pitch": 0.0}
    return _pack("random%d" % n_tris, vbo, idx, cam, cols, rows, 1.0, scene)


def upload(renderer, sc, ramp=DEFAULT_RAMP, mode_on=True, mode_radius=2, mode_thresh=12):
    renderer.set_mesh(sc["vbo"], sc["idx"])
    renderer.set_mesh_uv(sc.get("uv8"))        # set_mesh dropped any earlier coordinates; the atlas stays until replaced
    renderer.set_texture(sc.get("atlas"))
    renderer.set_camera(sc["proj"], sc["view"])
    renderer.set_model(sc.get("model"))
    renderer.set_lights(sc["lights"])
    renderer.set_ascii(ramp, mode_on, mode_radius, mode_thresh)


def algorithmic_bytes(n_tris, cols, rows):
    """Bytes one frame must move at minimum: the mesh as handed over (3 x 36 B vertices + 12 B of
    indices per triangle) read once, and RGBA8 + binary32 depth + ASCII written once per cell."""
    return n_tris * 120 + cols * rows * 9
