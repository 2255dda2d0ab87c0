"""ctypes wrapper around the CPU oracle (oracle/cpu_raster.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg as the checker / reported baseline.  The product path never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcpu_raster.so")
_lib = None


def build(force=False):
    """Compile the C oracle with the recipe in oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cpu_raster.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libcpu_raster.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.orc_matmul_pv.argtypes = [fp, fp, fp]
        L.orc_matmul_pv.restype = None
        L.orc_render.argtypes = [fp, ctypes.c_uint32, u32p, ctypes.c_uint32, fp, fp, fp, fp, fp,
                                 ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, u8p, fp, u32p]
        L.orc_render.restype = ctypes.c_int
        L.orc_render_textured.argtypes = L.orc_render.argtypes + [ctypes.POINTER(ctypes.c_uint16), u8p, ctypes.c_int, ctypes.c_int]
        L.orc_render_textured.restype = ctypes.c_int
        L.orc_render_model.argtypes = L.orc_render.argtypes[:6] + [fp] + L.orc_render.argtypes[6:] + [ctypes.POINTER(ctypes.c_uint16), u8p, ctypes.c_int, ctypes.c_int]
        L.orc_render_model.restype = ctypes.c_int
        L.orc_quantize.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p]
        L.orc_quantize.restype = ctypes.c_int
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def matmul_pv(proj, view):
    p, pp = _f(proj)
    v, vp = _f(view)
    out = np.zeros(16, dtype=np.float32)
    lib().orc_matmul_pv(pp, vp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def render(vbo9, idx, proj, view, lights, cols, rows, want_ids=True, uv8=None, atlas=None, model=None):
    """Returns (rgba[rows,cols,4] u8, depth[rows,cols] f32, ids[rows,cols] u32), all top-down.

    lights: dict with 'ambient', 'dir', 'dir_color' (3 floats each) and 'pl_pos', 'pl_col'
    (n x 3 each, n <= 8).
    uv8: optional uint16 [n_tris, 8] (u0, v0, u1, v1, u2, v2, flags, 0); atlas: optional uint8 [h, w, 4]
    as handed to texImage2D.  Both given: atlas texturing.
    model: optional column-major 4x4 model matrix (the vertex shader then has a uModel uniform).
    """
    vbo, vbop = _f(np.asarray(vbo9).reshape(-1))
    idx = np.ascontiguousarray(np.asarray(idx).reshape(-1), dtype=np.uint32)
    n_verts = vbo.size // 9
    n_tris = idx.size // 3
    _p, pp = _f(proj)
    _v, vp = _f(view)
    _a, ap = _f(lights["ambient"])
    _d, dp = _f(lights["dir"])
    _c, cp = _f(lights["dir_color"])
    plpos = np.asarray(lights.get("pl_pos", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
    plcol = np.asarray(lights.get("pl_col", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
    n_pl = plpos.shape[0]
    _pp, ppp = _f(plpos if n_pl else np.zeros(3))
    _pc, pcp = _f(plcol if n_pl else np.zeros(3))
    rgba = np.zeros((rows, cols, 4), dtype=np.uint8)
    depth = np.zeros((rows, cols), dtype=np.float32)
    ids = np.zeros((rows, cols), dtype=np.uint32) if want_ids else None
    args = [vbop, n_verts, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n_tris, pp, vp, ap, dp, cp,
            n_pl, ppp, pcp, cols, rows,
            rgba.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
            depth.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            ids.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if want_ids else None]
    tex = [None, None, 0, 0]
    if uv8 is not None and atlas is not None:
        uv8 = np.ascontiguousarray(uv8, dtype=np.uint16).reshape(-1, 8)
        atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
        if uv8.shape[0] != n_tris or atlas.ndim != 3 or atlas.shape[2] != 4:
            raise ValueError("uv8 must be [n_tris, 8] and atlas [h, w, 4]")
        tex = [uv8.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), atlas.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
               atlas.shape[1], atlas.shape[0]]
    if model is not None:
        _m, mp = _f(np.asarray(model).reshape(-1))
        if _m.size != 16:
            raise ValueError("model must have 16 elements")
        rc = lib().orc_render_model(*(args[:6] + [mp] + args[6:] + tex))
    elif tex[0] is not None:
        rc = lib().orc_render_textured(*(args + tex))
    else:
        rc = lib().orc_render(*args)
    if rc != 0:
        raise ValueError("orc_render rejected its arguments")
    return rgba, depth, ids


def quantize(rgba, cols, rows, ramp, mode_on=0, mode_radius=2, mode_thresh=12):
    """RGBA8 grid -> uint8[rows, cols] ASCII codes (override / ramp / optional modal smoothing)."""
    rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
    if rgba.size != cols * rows * 4:
        raise ValueError("rgba size does not match cols*rows*4")
    ramp_b = np.frombuffer(ramp.encode("latin-1") if isinstance(ramp, str) else bytes(ramp), dtype=np.uint8).copy()
    out = np.zeros((rows, cols), dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    rc = lib().orc_quantize(rgba.ctypes.data_as(u8p), cols, rows, ramp_b.ctypes.data_as(u8p), ramp_b.size,
                            int(mode_on), int(mode_radius), int(mode_thresh), out.ctypes.data_as(u8p))
    if rc != 0:
        raise ValueError("orc_quantize rejected its arguments")
    return out


def quantize_colors(rgba, cols, rows, ramp_len, steps=0, grayscale=False):
    """RGBA8 grid -> uint32[rows, cols] packed text colours of the DOM path (js/renderer.js:74-84)."""
    rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
    if rgba.size != cols * rows * 4:
        raise ValueError("rgba size does not match cols*rows*4")
    out = np.zeros((rows, cols), dtype=np.uint32)
    L = lib()
    L.orc_quantize_colors.restype = ctypes.c_int
    rc = L.orc_quantize_colors(rgba.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(cols), ctypes.c_int(rows),
                               ctypes.c_int(int(ramp_len)), ctypes.c_int(int(steps)), ctypes.c_int(int(bool(grayscale))),
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    if rc != 0:
        raise ValueError("orc_quantize_colors rejected its arguments")
    return out


def present(rgba, codes, cols, rows, atlas, cell_w, cell_h, gamma=1.0, grayscale=False, transparent_bg=False):
    """Glyph compositing: uint8[rows*cell_h, cols*cell_w, 4] image of the ASCII grid (ascii_pass_shader.js:190-230)."""
    rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
    codes = np.ascontiguousarray(np.asarray(codes, dtype=np.uint8).reshape(-1))
    atlas = np.ascontiguousarray(np.asarray(atlas, dtype=np.uint8).reshape(-1))
    if rgba.size != cols * rows * 4 or codes.size != cols * rows or atlas.size != 256 * (cell_w + 4) * (cell_h + 4):
        raise ValueError("present: array sizes do not match the grid / atlas layout")
    out = np.zeros((rows * cell_h, cols * cell_w, 4), dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L = lib()
    L.orc_present.restype = ctypes.c_int
    rc = L.orc_present(rgba.ctypes.data_as(u8p), codes.ctypes.data_as(u8p), ctypes.c_int(cols), ctypes.c_int(rows),
                       atlas.ctypes.data_as(u8p), ctypes.c_int(cell_w), ctypes.c_int(cell_h), ctypes.c_double(gamma),
                       ctypes.c_int(int(bool(grayscale))), ctypes.c_int(int(bool(transparent_bg))), out.ctypes.data_as(u8p))
    if rc != 0:
        raise ValueError("orc_present rejected its arguments")
    return out


class _RtScene(ctypes.Structure):
    _f, _i = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    _fields_ = [("n_spheres", ctypes.c_int), ("spheres4", _f), ("sphere_mat", _i),
                ("n_planes", ctypes.c_int), ("planes4", _f), ("plane_mat", _i),
                ("n_tris", ctypes.c_int), ("tri_a", _f), ("tri_b", _f), ("tri_c", _f), ("tri_mat", _i),
                ("n_mats", ctypes.c_int), ("mat_albedo", _f), ("mat_reflective", _i),
                ("n_pl", ctypes.c_int), ("pl_pos", _f), ("pl_col", _f),
                ("n_dl", ctypes.c_int), ("dl_dir", _f), ("dl_col", _f),
                ("env", ctypes.c_float * 3)]


def rt_render(u, cols, rows, cam_pos, yaw, pitch, fovy, pix_aspect=1.0):
    """Ray backend frame from the uniform arrays of the reference's RaytraceBackend (dict keyed by uniform
    name: uNumS, uSpheres, uSm, ... uEnv; absent arrays are empty) -> rgba uint8 [rows, cols, 4], top-down."""
    L = lib()
    fpt, ipt = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32)
    keep, sc = [], _RtScene()

    def arr(key, dtype, ptr):
        a = np.ascontiguousarray(u.get(key, np.zeros(0)), dtype=dtype).reshape(-1)
        keep.append(a)
        return a.ctypes.data_as(ptr) if a.size else None

    sc.n_spheres, sc.spheres4, sc.sphere_mat = int(u.get("uNumS", 0)), arr("uSpheres", np.float32, fpt), arr("uSm", np.int32, ipt)
    sc.n_planes, sc.planes4, sc.plane_mat = int(u.get("uNumP", 0)), arr("uPlanes", np.float32, fpt), arr("uPm", np.int32, ipt)
    sc.n_tris, sc.tri_a, sc.tri_b, sc.tri_c = int(u.get("uNumT", 0)), arr("uTa", np.float32, fpt), arr("uTb", np.float32, fpt), arr("uTc", np.float32, fpt)
    sc.tri_mat = arr("uTm", np.int32, ipt)
    sc.n_mats, sc.mat_albedo, sc.mat_reflective = int(u.get("uNumMats", 0)), arr("uMatAlbedo", np.float32, fpt), arr("uMatReflective", np.int32, ipt)
    sc.n_pl, sc.pl_pos, sc.pl_col = int(u.get("uNumPL", 0)), arr("uPLPos", np.float32, fpt), arr("uPLCol", np.float32, fpt)
    sc.n_dl, sc.dl_dir, sc.dl_col = int(u.get("uNumDL", 0)), arr("uDLDir", np.float32, fpt), arr("uDLCol", np.float32, fpt)
    env = np.asarray(u.get("uEnv", (0, 0, 0)), dtype=np.float32).reshape(-1)
    sc.env = (ctypes.c_float * 3)(*[float(x) for x in env[:3]])
    L.orc_rt_render.argtypes = [ctypes.POINTER(_RtScene), ctypes.c_int, ctypes.c_int, fpt, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_uint8)]
    L.orc_rt_render.restype = ctypes.c_int
    cam, camp = _f(cam_pos)
    rgba = np.zeros((rows, cols, 4), dtype=np.uint8)
    if L.orc_rt_render(ctypes.byref(sc), cols, rows, camp, yaw, pitch, fovy, pix_aspect, rgba.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))) != 0:
        raise ValueError("orc_rt_render rejected its arguments")
    return rgba


def rt_camera(yaw, pitch, fovy):
    """(uu, vv, ww, focal) of the ray backend's camera."""
    L = lib()
    L.orc_rt_camera.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
    L.orc_rt_camera.restype = None
    b = np.zeros(10, dtype=np.float32)
    L.orc_rt_camera(yaw, pitch, fovy, b.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return b[0:3], b[3:6], b[6:9], float(b[9])
