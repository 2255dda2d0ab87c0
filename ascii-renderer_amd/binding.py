"""ctypes binding to the C ABI of libascii_rast.so (include/ascii_rast.h).

There is no CPU fallback: if the library is missing or no GPU is present, construction raises.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libascii_rast.so")
N_KERNELS = 12
EXPORTS = [
    "ar_create", "ar_destroy", "ar_last_error", "ar_set_mesh", "ar_set_camera", "ar_set_lights",
    "ar_set_ascii", "ar_set_strip", "ar_set_id_output", "ar_render", "ar_render_async", "ar_sync", "ar_read",
    "ar_device_buffer", "ar_stream", "ar_host_alloc", "ar_host_free", "ar_stream_wait_frame", "ar_frame_wait_stream", "ar_set_stream",
    "ar_set_ascii_target", "ar_quantize", "ar_quantize_colors", "ar_read_colors", "ar_present", "ar_set_profiling", "ar_kernel_stats",
    "ar_frame_counters", "ar_set_option", "ar_set_mesh_uv", "ar_set_texture", "ar_set_mesh_indexed", "ar_read_mesh",
    "ar_rt_set_scene", "ar_rt_render", "ar_rt_render_async", "ar_quantize_static", "ar_set_model",
    "ar_read_async", "ar_wait", "ar_poll", "ar_update_positions",
    "ar_render_batch_async", "ar_read_batch", "ar_batch_device_buffer",
    "ar_strip_rows", "ar_group_create", "ar_group_destroy", "ar_group_last_error", "ar_group_size", "ar_group_ctx",
    "ar_group_render_async", "ar_group_sync", "ar_group_frames_gathered", "ar_group_read",
]
_lib = None

_fp = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)


class RtScene(ctypes.Structure):
    """ar_rt_scene of include/ascii_rast.h."""
    _fields_ = [("n_spheres", ctypes.c_int), ("spheres4", _fp), ("sphere_mat", _i32p),
                ("n_planes", ctypes.c_int), ("planes4", _fp), ("plane_mat", _i32p),
                ("n_tris", ctypes.c_int), ("tri_a", _fp), ("tri_b", _fp), ("tri_c", _fp), ("tri_mat", _i32p),
                ("n_mats", ctypes.c_int), ("mat_albedo", _fp), ("mat_reflective", _i32p),
                ("n_pl", ctypes.c_int), ("pl_pos", _fp), ("pl_col", _fp),
                ("n_dl", ctypes.c_int), ("dl_dir", _fp), ("dl_col", _fp),
                ("env", ctypes.c_float * 3)]


class AsciiRastError(RuntimeError):
    pass


def load_library():
    """Loads libascii_rast.so (built by ascii-renderer_amd/csrc/Makefile) and declares its signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AsciiRastError(
            "HIP library not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ascii-renderer_amd/csrc`)" % LIB_PATH)
    if "torch" not in sys.modules:
        # PyTorch's wheel carries its own HIP runtime under another file name.  The loader shares it with this
        # library only when PyTorch is loaded first; the other way round the process ends up with two runtimes
        # and PyTorch then finds no device.  The harness uses both, so PyTorch (when installed) goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    L.ar_create.argtypes = [ctypes.c_int]
    L.ar_create.restype = vp
    L.ar_destroy.argtypes = [vp]
    L.ar_destroy.restype = None
    L.ar_last_error.argtypes = [vp]
    L.ar_last_error.restype = ctypes.c_char_p
    L.ar_set_mesh.argtypes = [vp, _fp, ctypes.c_uint32, _u32p, ctypes.c_uint32]
    L.ar_rt_set_scene.argtypes = [vp, ctypes.POINTER(RtScene)]
    L.ar_rt_render.argtypes = [vp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _u8p, _u8p]
    L.ar_rt_render_async.argtypes = [vp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    L.ar_set_mesh_indexed.argtypes = [vp, _fp, ctypes.c_uint32, _u32p, ctypes.c_uint32, _fp]
    L.ar_read_mesh.argtypes = [vp, _fp, _u32p, _u32p, _u32p]
    L.ar_update_positions.argtypes = [vp, _fp, ctypes.c_uint32, ctypes.c_int, ctypes.c_int]
    L.ar_update_positions.restype = ctypes.c_int
    L.ar_set_mesh_uv.argtypes = [vp, ctypes.POINTER(ctypes.c_uint16), ctypes.c_uint32]
    L.ar_set_texture.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int]
    L.ar_set_camera.argtypes = [vp, _fp, _fp]
    L.ar_set_model.argtypes = [vp, _fp]
    L.ar_set_model.restype = ctypes.c_int
    L.ar_set_lights.argtypes = [vp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp]
    L.ar_set_ascii.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.ar_set_strip.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.ar_set_id_output.argtypes = [vp, ctypes.c_int]
    L.ar_render.argtypes = [vp, ctypes.c_int, ctypes.c_int, _u8p, _fp, _u8p]
    L.ar_render_async.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.ar_sync.argtypes = [vp]
    L.ar_read.argtypes = [vp, _u8p, _fp, _u8p, _u32p]
    L.ar_read_async.argtypes = [vp, _u8p, _fp, _u8p, _u32p, ctypes.POINTER(ctypes.c_uint64)]
    L.ar_wait.argtypes = [vp, ctypes.c_uint64]
    L.ar_poll.argtypes = [vp, ctypes.c_uint64]
    for name in ("ar_read_async", "ar_wait", "ar_poll"):
        getattr(L, name).restype = ctypes.c_int
    _ip = ctypes.POINTER(ctypes.c_int)
    L.ar_strip_rows.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, _ip]
    L.ar_group_create.argtypes = [ctypes.c_int, _ip, ctypes.c_int]
    L.ar_group_create.restype = vp
    L.ar_group_destroy.argtypes = [vp]
    L.ar_group_destroy.restype = None
    L.ar_group_last_error.argtypes = [vp]
    L.ar_group_last_error.restype = ctypes.c_char_p
    L.ar_group_size.argtypes = [vp]
    L.ar_group_ctx.argtypes = [vp, ctypes.c_int]
    L.ar_group_ctx.restype = vp
    L.ar_group_render_async.argtypes = [vp, ctypes.c_int, ctypes.c_int, _fp, _fp]
    L.ar_group_sync.argtypes = [vp]
    L.ar_group_frames_gathered.argtypes = [vp]
    L.ar_group_read.argtypes = [vp, ctypes.c_int, ctypes.c_int, _u8p]
    for name in ("ar_strip_rows", "ar_group_size", "ar_group_render_async", "ar_group_sync", "ar_group_frames_gathered", "ar_group_read"):
        getattr(L, name).restype = ctypes.c_int
    L.ar_render_batch_async.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_size_t]
    L.ar_render_batch_async.restype = ctypes.c_int
    L.ar_read_batch.argtypes = [vp, ctypes.c_int, _u8p, _fp, _u8p, _u32p]
    L.ar_read_batch.restype = ctypes.c_int
    L.ar_batch_device_buffer.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.ar_batch_device_buffer.restype = vp
    L.ar_device_buffer.argtypes = [vp, ctypes.c_int]
    L.ar_device_buffer.restype = vp
    L.ar_stream.argtypes = [vp]
    L.ar_stream.restype = vp
    L.ar_host_alloc.argtypes = [vp, ctypes.c_size_t]
    L.ar_host_alloc.restype = vp
    L.ar_host_free.argtypes = [vp]
    L.ar_host_free.restype = None
    L.ar_set_stream.argtypes = [vp, vp]
    L.ar_set_ascii_target.argtypes = [vp, vp, ctypes.c_int]
    L.ar_stream_wait_frame.argtypes = [vp, vp]
    L.ar_frame_wait_stream.argtypes = [vp, vp]
    L.ar_quantize.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int, _u8p]
    L.ar_quantize_static.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int, _u8p]
    L.ar_quantize_static.restype = ctypes.c_int
    L.ar_quantize_colors.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u32p]
    L.ar_read_colors.argtypes = [vp, ctypes.c_int, ctypes.c_int, _u32p]
    L.ar_present.argtypes = [vp, _u8p, ctypes.c_int, ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                             ctypes.c_int, ctypes.c_int, _u8p]
    L.ar_set_profiling.argtypes = [vp, ctypes.c_int]
    L.ar_kernel_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), _u32p, _fp]
    L.ar_frame_counters.argtypes = [vp, _u32p]
    L.ar_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_longlong]
    for name in ("ar_set_id_output", "ar_set_mesh", "ar_set_camera", "ar_set_lights", "ar_set_ascii", "ar_set_strip", "ar_set_id_output", "ar_render",
                 "ar_render_async", "ar_sync", "ar_read", "ar_quantize", "ar_stream_wait_frame", "ar_frame_wait_stream", "ar_set_stream", "ar_set_ascii_target", "ar_quantize_colors", "ar_read_colors", "ar_present", "ar_set_profiling", "ar_kernel_stats",
                 "ar_frame_counters", "ar_set_option", "ar_set_mesh_uv", "ar_set_texture", "ar_set_mesh_indexed", "ar_read_mesh", "ar_rt_set_scene", "ar_rt_render",
                 "ar_rt_render_async"):
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def strip_rows(rows, n_strips, strip):
    """(row0, row1) of one strip when `rows` rows are cut into n_strips (ar_strip_rows: plain arithmetic, no GPU)."""
    L = load_library()
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    if L.ar_strip_rows(int(rows), int(n_strips), int(strip), ctypes.byref(a), ctypes.byref(b)):
        raise ValueError("bad strip %d of %d over %d rows" % (strip, n_strips, rows))
    return a.value, b.value


class Renderer:
    """One GPU renderer context (ar_ctx).  Outputs are numpy arrays, row 0 = top of the image."""

    def __init__(self, device=0, _borrowed=None):
        self._L = load_library()
        self._owned = _borrowed is None
        self._h = self._L.ar_create(int(device)) if self._owned else _borrowed
        if not self._h:
            raise AsciiRastError("ar_create failed: " + (self._L.ar_last_error(None) or b"").decode())
        self.cols = self.rows = 0

    def _check(self, rc):
        if rc != 0:
            raise AsciiRastError((self._L.ar_last_error(self._h) or b"").decode() + " (code %d)" % rc)

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.ar_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mesh(self, vbo9, idx):
        vbo, vp = _f32(np.asarray(vbo9).reshape(-1))
        idx = np.ascontiguousarray(np.asarray(idx).reshape(-1), dtype=np.uint32)
        if vbo.size % 9 or idx.size % 3:
            raise ValueError("vbo must hold 9 floats per vertex and idx 3 indices per triangle")
        self._check(self._L.ar_set_mesh(self._h, vp, vbo.size // 9, idx.ctypes.data_as(_u32p), idx.size // 3))

    def set_mesh_indexed(self, positions, indices, colors=None):
        """Indexed triangle list (positions [V, 3] f32, indices [T, 3] u32, optional per-triangle albedo [T, 3]);
        the unshared flat-shaded vertex buffer is built on the GPU."""
        pos, pp = _f32(np.asarray(positions).reshape(-1))
        ind = np.ascontiguousarray(np.asarray(indices).reshape(-1), dtype=np.uint32)
        if pos.size % 3 or ind.size % 3:
            raise ValueError("positions need 3 floats per vertex and indices 3 per triangle")
        cp = None
        if colors is not None:
            col, cp = _f32(np.asarray(colors).reshape(-1))
            if col.size != ind.size:
                raise ValueError("colors must hold 3 floats per triangle")
        self._check(self._L.ar_set_mesh_indexed(self._h, pp, pos.size // 3, ind.ctypes.data_as(_u32p), ind.size // 3, cp))

    def update_positions(self, data, flat_normals=False, resort=False):
        """New vertex data for the current mesh in place: data is [V, 9] (whole vertices) or [V, 3] (positions only)."""
        a = np.ascontiguousarray(data, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] not in (3, 9):
            raise ValueError("data must be [n_verts, 3] or [n_verts, 9]")
        self._check(self._L.ar_update_positions(self._h, a.ctypes.data_as(_fp), a.shape[0], a.shape[1],
                                                (1 if flat_normals else 0) | (2 if resort else 0)))

    def read_mesh(self):
        """(vbo float32 [V, 9], idx uint32 [3 T]) of the current mesh as held on the GPU."""
        nv, nt = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self._check(self._L.ar_read_mesh(self._h, None, None, ctypes.byref(nv), ctypes.byref(nt)))
        vbo = np.zeros((nv.value, 9), dtype=np.float32)
        idx = np.zeros(3 * nt.value, dtype=np.uint32)
        if nt.value:
            self._check(self._L.ar_read_mesh(self._h, vbo.ctypes.data_as(_fp), idx.ctypes.data_as(_u32p), None, None))
        return vbo, idx

    def rt_set_scene(self, u):
        """Ray backend scene from the uniform arrays RaytraceBackend uploads (raytrace.js:351-389): dict with
        uNumS, uSpheres (float [n, 4]), uSm (int), uNumP, uPlanes, uPm, uNumT, uTa, uTb, uTc, uTm, uNumMats,
        uMatAlbedo, uMatReflective, uNumPL, uPLPos, uPLCol, uNumDL, uDLDir, uDLCol, uEnv; absent arrays count as empty."""
        keep, sc = [], RtScene()

        def arr(key, dtype, ptr):
            a = np.ascontiguousarray(u.get(key, np.zeros(0)), dtype=dtype).reshape(-1)
            keep.append(a)
            return a.ctypes.data_as(ptr) if a.size else None

        sc.n_spheres, sc.spheres4, sc.sphere_mat = int(u.get("uNumS", 0)), arr("uSpheres", np.float32, _fp), arr("uSm", np.int32, _i32p)
        sc.n_planes, sc.planes4, sc.plane_mat = int(u.get("uNumP", 0)), arr("uPlanes", np.float32, _fp), arr("uPm", np.int32, _i32p)
        sc.n_tris, sc.tri_a, sc.tri_b, sc.tri_c = int(u.get("uNumT", 0)), arr("uTa", np.float32, _fp), arr("uTb", np.float32, _fp), arr("uTc", np.float32, _fp)
        sc.tri_mat = arr("uTm", np.int32, _i32p)
        sc.n_mats, sc.mat_albedo, sc.mat_reflective = int(u.get("uNumMats", 0)), arr("uMatAlbedo", np.float32, _fp), arr("uMatReflective", np.int32, _i32p)
        sc.n_pl, sc.pl_pos, sc.pl_col = int(u.get("uNumPL", 0)), arr("uPLPos", np.float32, _fp), arr("uPLCol", np.float32, _fp)
        sc.n_dl, sc.dl_dir, sc.dl_col = int(u.get("uNumDL", 0)), arr("uDLDir", np.float32, _fp), arr("uDLCol", np.float32, _fp)
        env = np.asarray(u.get("uEnv", (0, 0, 0)), dtype=np.float32).reshape(-1)
        sc.env = (ctypes.c_float * 3)(*[float(x) for x in env[:3]])
        self._check(self._L.ar_rt_set_scene(self._h, ctypes.byref(sc)))

    def rt_render(self, cols, rows, cam_pos, yaw, pitch, fovy, pix_aspect=1.0):
        """One frame of the ray backend -> dict(rgba [rows, cols, 4], ascii [rows, cols])."""
        _c, cp = _f32(cam_pos)
        rgba = np.zeros((rows, cols, 4), dtype=np.uint8)
        ascii_ = np.zeros((rows, cols), dtype=np.uint8)
        self._check(self._L.ar_rt_render(self._h, cols, rows, cp, yaw, pitch, fovy, pix_aspect,
                                         rgba.ctypes.data_as(_u8p), ascii_.ctypes.data_as(_u8p)))
        return {"rgba": rgba, "ascii": ascii_}

    def rt_render_async(self, cols, rows, cam_pos, yaw, pitch, fovy, pix_aspect=1.0):
        _c, cp = _f32(cam_pos)
        self._check(self._L.ar_rt_render_async(self._h, cols, rows, cp, yaw, pitch, fovy, pix_aspect))
        self.cols, self.rows = cols, rows

    def set_mesh_uv(self, uv8):
        """uv8: uint16 [n_tris, 8] rows (u0, v0, u1, v1, u2, v2, flags, 0) for the current mesh, or None to remove."""
        if uv8 is None:
            self._check(self._L.ar_set_mesh_uv(self._h, None, 0))
            return
        uv8 = np.ascontiguousarray(uv8, dtype=np.uint16)
        if uv8.ndim != 2 or uv8.shape[1] != 8:
            raise ValueError("uv8 must be [n_tris, 8]")
        self._check(self._L.ar_set_mesh_uv(self._h, uv8.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), uv8.shape[0]))

    def set_texture(self, atlas):
        """atlas: uint8 [height, width, 4] as the reference hands it to texImage2D, or None to remove."""
        if atlas is None:
            self._check(self._L.ar_set_texture(self._h, None, 0, 0))
            return
        atlas = np.ascontiguousarray(atlas, dtype=np.uint8)
        if atlas.ndim != 3 or atlas.shape[2] != 4:
            raise ValueError("atlas must be [height, width, 4]")
        self._check(self._L.ar_set_texture(self._h, atlas.ctypes.data_as(_u8p), atlas.shape[1], atlas.shape[0]))

    def set_camera(self, proj, view):
        _p, pp = _f32(proj)
        _v, vp = _f32(view)
        if _p.size != 16 or _v.size != 16:
            raise ValueError("proj and view must have 16 elements")
        self._check(self._L.ar_set_camera(self._h, pp, vp))

    def set_model(self, model):
        """Column-major 4x4 model matrix applied to the current mesh on the GPU; None removes it."""
        if model is None:
            self._check(self._L.ar_set_model(self._h, None))
            return
        _m, mp = _f32(np.asarray(model).reshape(-1))
        if _m.size != 16:
            raise ValueError("model must have 16 elements")
        self._check(self._L.ar_set_model(self._h, mp))

    @staticmethod
    def prepare_camera(proj, view):
        """The two matrices as ready-made C arrays for set_camera_prepared (a per-frame camera without per-frame
        conversions on the host)."""
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(-1)
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(-1)
        if p.size != 16 or v.size != 16:
            raise ValueError("proj and view must have 16 elements")
        return (ctypes.c_float * 16)(*p.tolist()), (ctypes.c_float * 16)(*v.tolist())

    def set_camera_prepared(self, prepared):
        rc = self._L.ar_set_camera(self._h, prepared[0], prepared[1])
        if rc:
            self._check(rc)

    def set_lights(self, lights):
        _a, ap = _f32(lights["ambient"])
        _d, dp = _f32(lights["dir"])
        _c, cp = _f32(lights["dir_color"])
        plpos = np.asarray(lights.get("pl_pos", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
        plcol = np.asarray(lights.get("pl_col", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
        n = plpos.shape[0]
        _pp, ppp = _f32(plpos if n else np.zeros(3))
        _pc, pcp = _f32(plcol if n else np.zeros(3))
        self._check(self._L.ar_set_lights(self._h, ap, dp, cp, n, ppp, pcp))

    def set_ascii(self, ramp="@%#*+=-:. ", mode_on=True, mode_radius=2, mode_thresh=12):
        rb = np.frombuffer(ramp.encode("latin-1") if isinstance(ramp, str) else bytes(ramp), dtype=np.uint8).copy()
        self._check(self._L.ar_set_ascii(self._h, rb.ctypes.data_as(_u8p), rb.size, int(bool(mode_on)),
                                         int(mode_radius), int(mode_thresh)))

    def set_strip(self, row0, row1):
        self._check(self._L.ar_set_strip(self._h, int(row0), int(row1)))

    def set_id_output(self, on):
        self._check(self._L.ar_set_id_output(self._h, int(bool(on))))

    def render(self, cols, rows, want_ids=False):
        """Blocking frame; returns dict(rgba, depth, ascii[, ids])."""
        self._check(self._L.ar_set_id_output(self._h, int(bool(want_ids))))
        self.render_async(cols, rows)
        return self.read(want_ids=want_ids)

    def render_async(self, cols, rows):
        self._check(self._L.ar_render_async(self._h, int(cols), int(rows)))
        self.cols, self.rows = int(cols), int(rows)

    def sync(self):
        self._check(self._L.ar_sync(self._h))

    def read(self, want_ids=False, rgba=True, depth=True, ascii=True):
        r, c = self.rows, self.cols
        out = {}
        a_rgba = np.zeros((r, c, 4), dtype=np.uint8) if rgba else None
        a_depth = np.zeros((r, c), dtype=np.float32) if depth else None
        a_ascii = np.zeros((r, c), dtype=np.uint8) if ascii else None
        a_ids = np.zeros((r, c), dtype=np.uint32) if want_ids else None
        self._check(self._L.ar_read(
            self._h,
            a_rgba.ctypes.data_as(_u8p) if rgba else None,
            a_depth.ctypes.data_as(_fp) if depth else None,
            a_ascii.ctypes.data_as(_u8p) if ascii else None,
            a_ids.ctypes.data_as(_u32p) if want_ids else None))
        out["rgba"], out["depth"], out["ascii"], out["ids"] = a_rgba, a_depth, a_ascii, a_ids
        return out

    def render_into(self, cols, rows, rgba=None, depth=None, ascii=None):
        """ar_render with caller-owned host buffers (the backend's render(time, framebuffer, ...) shape)."""
        self._check(self._L.ar_render(
            self._h, int(cols), int(rows),
            rgba.ctypes.data_as(_u8p) if rgba is not None else None,
            depth.ctypes.data_as(_fp) if depth is not None else None,
            ascii.ctypes.data_as(_u8p) if ascii is not None else None))
        self.cols, self.rows = int(cols), int(rows)

    @staticmethod
    def pack_cameras(cameras):
        """[(proj, view), ...] -> float32 [n, 32] as ar_render_batch_async takes it."""
        out = np.zeros((len(cameras), 32), dtype=np.float32)
        for k, (proj, view) in enumerate(cameras):
            out[k, :16] = np.asarray(proj, dtype=np.float32).reshape(-1)
            out[k, 16:] = np.asarray(view, dtype=np.float32).reshape(-1)
        return out

    def render_batch_async(self, cols, rows, packed_cameras, ascii_stride=0):
        """len(packed_cameras) frames in one chain of launches (packed_cameras: pack_cameras' array)."""
        cams = packed_cameras
        rc = self._L.ar_render_batch_async(self._h, int(cols), int(rows), int(cams.shape[0]), cams.ctypes.data_as(_fp), int(ascii_stride))
        if rc:
            self._check(rc)
        self.cols, self.rows = int(cols), int(rows)

    def read_batch(self, frame, want_ids=False, rgba=True, depth=True, ascii=True):
        r, c = self.rows, self.cols
        a_rgba = np.zeros((r, c, 4), dtype=np.uint8) if rgba else None
        a_depth = np.zeros((r, c), dtype=np.float32) if depth else None
        a_ascii = np.zeros((r, c), dtype=np.uint8) if ascii else None
        a_ids = np.zeros((r, c), dtype=np.uint32) if want_ids else None
        self._check(self._L.ar_read_batch(
            self._h, int(frame),
            a_rgba.ctypes.data_as(_u8p) if rgba else None, a_depth.ctypes.data_as(_fp) if depth else None,
            a_ascii.ctypes.data_as(_u8p) if ascii else None, a_ids.ctypes.data_as(_u32p) if want_ids else None))
        return {"rgba": a_rgba, "depth": a_depth, "ascii": a_ascii, "ids": a_ids}

    def batch_device_buffer(self, which, frame):
        return self._L.ar_batch_device_buffer(self._h, {"rgba": 0, "depth": 1, "ascii": 2}[which], int(frame))

    def read_async(self, rgba=None, depth=None, ascii=None, ids=None):
        """Queues copies of the last enqueued frame's planes into the given host arrays (page-locked ones from
        host_array for a copy that does not block) behind that frame and returns a ticket for wait() / poll()."""
        t = ctypes.c_uint64(0)
        self._check(self._L.ar_read_async(
            self._h,
            rgba.ctypes.data_as(_u8p) if rgba is not None else None,
            depth.ctypes.data_as(_fp) if depth is not None else None,
            ascii.ctypes.data_as(_u8p) if ascii is not None else None,
            ids.ctypes.data_as(_u32p) if ids is not None else None, ctypes.byref(t)))
        return t.value

    def wait(self, ticket):
        self._check(self._L.ar_wait(self._h, ctypes.c_uint64(int(ticket))))

    def poll(self, ticket):
        rc = self._L.ar_poll(self._h, ctypes.c_uint64(int(ticket)))
        if rc < 0:
            self._check(rc)
        return rc == 1

    def device_buffer(self, which):
        return self._L.ar_device_buffer(self._h, {"rgba": 0, "depth": 1, "ascii": 2}[which])

    def host_array(self, shape, dtype):
        """numpy array over page-locked host memory (ar_host_alloc): device->host copies into it run at
        the full PCIe rate.  The memory is released when the array (and its views) are gone."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self._L.ar_host_alloc(self._h, n)
        if not p:
            raise AsciiRastError((self._L.ar_last_error(self._h) or b"").decode())
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(p)
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        import weakref
        weakref.finalize(buf, self._L.ar_host_free, p)
        return arr

    def set_stream(self, hip_stream):
        """Runs the following frames on the caller's stream (integer hipStream_t); 0 / None: own streams."""
        self._check(self._L.ar_set_stream(self._h, ctypes.c_void_p(int(hip_stream or 0))))

    def set_ascii_target(self, device_ptr, first_row=0):
        """The following frames write ASCII row r at device_ptr + (r - first_row) * cols; None: own buffer."""
        self._check(self._L.ar_set_ascii_target(self._h, ctypes.c_void_p(int(device_ptr or 0)), int(first_row)))

    def stream_wait_frame(self, hip_stream):
        """Work enqueued on hip_stream (an integer hipStream_t) from now on waits for the frames enqueued so far."""
        self._check(self._L.ar_stream_wait_frame(self._h, ctypes.c_void_p(int(hip_stream))))

    def frame_wait_stream(self, hip_stream):
        """Frames enqueued from now on write their outputs only after the work now on hip_stream is done."""
        self._check(self._L.ar_frame_wait_stream(self._h, ctypes.c_void_p(int(hip_stream))))

    def quantize(self, rgba, cols, rows):
        rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
        if rgba.size != cols * rows * 4:
            raise ValueError("rgba size does not match cols*rows*4")
        out = np.zeros((rows, cols), dtype=np.uint8)
        self._check(self._L.ar_quantize(self._h, rgba.ctypes.data_as(_u8p), int(cols), int(rows), out.ctypes.data_as(_u8p)))
        return out

    def quantize_into(self, rgba, cols, rows, out):
        """quantize() with caller-owned arrays (uint8, C-contiguous): no conversions or allocations on the way."""
        self._check(self._L.ar_quantize(self._h, rgba.ctypes.data_as(_u8p), int(cols), int(rows), out.ctypes.data_as(_u8p)))
        return out

    def quantize_static(self, rgba, cols, rows):
        """quantize() through the library's fixed staging array: no device allocation on the way."""
        rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
        if rgba.size != cols * rows * 4:
            raise ValueError("rgba size does not match cols*rows*4")
        out = np.zeros((rows, cols), dtype=np.uint8)
        self._check(self._L.ar_quantize_static(self._h, rgba.ctypes.data_as(_u8p), int(cols), int(rows), out.ctypes.data_as(_u8p)))
        return out

    def quantize_colors(self, rgba, cols, rows, steps=0, grayscale=False):
        """uint32[rows, cols] packed text colours of an RGBA8 grid (colour half of compositeAndQuantize)."""
        rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
        if rgba.size != cols * rows * 4:
            raise ValueError("rgba size does not match cols*rows*4")
        out = np.zeros((rows, cols), dtype=np.uint32)
        self._check(self._L.ar_quantize_colors(self._h, rgba.ctypes.data_as(_u8p), int(cols), int(rows), int(steps),
                                               int(bool(grayscale)), out.ctypes.data_as(_u32p)))
        return out

    def present(self, rgba, cols, rows, atlas, cell_w, cell_h, gamma=1.0, grayscale=False, transparent_bg=False, out=None):
        """uint8[rows*cell_h, cols*cell_w, 4] image of the ASCII grid drawn with the caller's glyph atlas."""
        rgba = np.ascontiguousarray(np.asarray(rgba, dtype=np.uint8).reshape(-1))
        atlas = np.ascontiguousarray(np.asarray(atlas, dtype=np.uint8).reshape(-1))
        if rgba.size != cols * rows * 4 or atlas.size != 256 * (cell_w + 4) * (cell_h + 4):
            raise ValueError("present: array sizes do not match the grid / atlas layout")
        if out is None:
            out = np.zeros((rows * cell_h, cols * cell_w, 4), dtype=np.uint8)
        self._check(self._L.ar_present(self._h, rgba.ctypes.data_as(_u8p), int(cols), int(rows), atlas.ctypes.data_as(_u8p),
                                       int(cell_w), int(cell_h), float(gamma), int(bool(grayscale)), int(bool(transparent_bg)),
                                       out.ctypes.data_as(_u8p)))
        return out

    def read_colors(self, steps=0, grayscale=False):
        """Packed text colours of the last rendered frame."""
        out = np.zeros((self.rows, self.cols), dtype=np.uint32)
        self._check(self._L.ar_read_colors(self._h, int(steps), int(bool(grayscale)), out.ctypes.data_as(_u32p)))
        return out

    def set_profiling(self, on):
        self._check(self._L.ar_set_profiling(self._h, int(bool(on))))

    def kernel_stats(self):
        """{kernel name: (launches, total ms)} since the last call (profiling must be on)."""
        names = (ctypes.c_char_p * N_KERNELS)()
        launches = np.zeros(N_KERNELS, dtype=np.uint32)
        ms = np.zeros(N_KERNELS, dtype=np.float32)
        self._check(self._L.ar_kernel_stats(self._h, names, launches.ctypes.data_as(_u32p), ms.ctypes.data_as(_fp)))
        return {names[k].decode(): (int(launches[k]), float(ms[k])) for k in range(N_KERNELS) if names[k]}

    def frame_counters(self):
        out = np.zeros(8, dtype=np.uint32)
        self._check(self._L.ar_frame_counters(self._h, out.ctypes.data_as(_u32p)))
        return {"triangles": int(out[0]), "medium_records": int(out[1]), "large_records": int(out[2]),
                "bin_records": int(out[3]), "pool_overflow": int(out[4]), "tiled": int(out[5])}

    def set_option(self, name, value):
        self._check(self._L.ar_set_option(self._h, {"path": 1, "pool_records": 2, "overlap": 3, "tile_rows": 4, "work_classes": 5, "fold_scan": 6, "smoothing_kernel": 7}[name], int(value)))


class Group:
    """Several GPUs in one process (ar_group): one frame cut into row strips, a context and a host thread per device,
    the ASCII strips combined by RCCL all-gathers every `batch` frames."""

    def __init__(self, device_ids, batch=16):
        self._L = load_library()
        ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._g = self._L.ar_group_create(len(device_ids), ids, int(batch))
        if not self._g:
            raise AsciiRastError("ar_group_create failed: " + (self._L.ar_group_last_error(None) or b"").decode())
        self.n, self.batch = len(device_ids), int(batch)
        self.members = [Renderer(_borrowed=self._L.ar_group_ctx(self._g, i)) for i in range(self.n)]
        self.cols = self.rows = 0

    def _check(self, rc):
        if rc != 0:
            raise AsciiRastError((self._L.ar_group_last_error(self._g) or b"").decode() + " (code %d)" % rc)

    def close(self):
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()
            self._L.ar_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_async(self, cols, rows, prepared_camera):
        """One frame on every member; prepared_camera as Renderer.prepare_camera returns it."""
        rc = self._L.ar_group_render_async(self._g, int(cols), int(rows), prepared_camera[0], prepared_camera[1])
        if rc:
            self._check(rc)
        self.cols, self.rows = int(cols), int(rows)

    def sync(self):
        self._check(self._L.ar_group_sync(self._g))

    def frames_gathered(self):
        return self._L.ar_group_frames_gathered(self._g)

    def read(self, frame=0, member=0):
        """uint8 [rows, cols]: frame `frame` of the last gathered batch as member `member` holds it."""
        out = np.zeros((self.rows, self.cols), dtype=np.uint8)
        self._check(self._L.ar_group_read(self._g, int(member), int(frame), out.ctypes.data_as(_u8p)))
        return out
