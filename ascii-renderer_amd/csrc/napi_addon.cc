// napi_addon.cc - thin N-API binding of the C ABI in include/ascii_rast.h for the JavaScript host.
//
// Every export is a one-to-one wrapper: typed arrays in, status out.  Setup calls throw a JS Error
// carrying ar_last_error(); nothing here computes anything.  The handle returned by create() is an
// external whose finalizer destroys the context if the host forgot to.
#define NAPI_VERSION 6
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "ascii_rast.h"

namespace {

#define NAPI_OK(env, call)                                                       \
    do {                                                                         \
        if ((call) != napi_ok) {                                                 \
            napi_throw_error(env, nullptr, "ascii_rast addon: N-API call failed: " #call); \
            return nullptr;                                                      \
        }                                                                        \
    } while (0)

struct Handle {
    ar_ctx *ctx;
    bool owned;  // false for a member of a group: the group destroys it
};

void finalize_handle(napi_env, void *data, void *) {
    Handle *h = static_cast<Handle *>(data);
    if (h->ctx && h->owned) ar_destroy(h->ctx);
    delete h;
}

struct GroupHandle {
    ar_group *group;
};

void finalize_group(napi_env, void *data, void *) {
    GroupHandle *h = static_cast<GroupHandle *>(data);
    if (h->group) ar_group_destroy(h->group);
    delete h;
}

napi_value throw_ar(napi_env env, ar_ctx *ctx, const char *what) {
    char msg[384];
    snprintf(msg, sizeof msg, "%s: %s", what, ar_last_error(ctx));
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_type_error(env, nullptr, "ascii_rast addon: missing arguments");
        return false;
    }
    return true;
}

ar_ctx *get_ctx(napi_env env, napi_value v) {
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !static_cast<Handle *>(p)->ctx) {
        napi_throw_type_error(env, nullptr, "ascii_rast addon: invalid or destroyed renderer handle");
        return nullptr;
    }
    return static_cast<Handle *>(p)->ctx;
}

// Typed array (or null/undefined -> nullptr, length 0).  want: napi_float32_array, napi_uint32_array,
// napi_uint8_array (Uint8ClampedArray is accepted where Uint8Array is).
bool get_typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *length, bool allow_null) {
    napi_valuetype vt;
    napi_typeof(env, v, &vt);
    if (vt == napi_null || vt == napi_undefined) {
        *data = nullptr; *length = 0;
        if (!allow_null) napi_throw_type_error(env, nullptr, "ascii_rast addon: typed array expected");
        return allow_null;
    }
    bool is_ta = false;
    napi_is_typedarray(env, v, &is_ta);
    if (!is_ta) { napi_throw_type_error(env, nullptr, "ascii_rast addon: typed array expected"); return false; }
    napi_typedarray_type type;
    napi_value buf;
    size_t off;
    if (napi_get_typedarray_info(env, v, &type, length, data, &buf, &off) != napi_ok) return false;
    const bool ok = type == want || (want == napi_uint8_array && type == napi_uint8_clamped_array);
    if (!ok) napi_throw_type_error(env, nullptr, "ascii_rast addon: typed array of the wrong element type");
    return ok;
}

int32_t get_i32(napi_env env, napi_value v) {
    int32_t x = 0;
    napi_get_value_int32(env, v, &x);
    return x;
}

napi_value js_create(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = ar_create(get_i32(env, argv[0]));
    if (!ctx) return throw_ar(env, nullptr, "ar_create");
    Handle *h = new Handle{ctx, true};
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, h, finalize_handle, nullptr, &ext));
    return ext;
}

napi_value js_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    void *p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        Handle *h = static_cast<Handle *>(p);
        if (h->ctx) { if (h->owned) ar_destroy(h->ctx); h->ctx = nullptr; }  // idempotent, like RasterBackend.dispose
    }
    return nullptr;
}

napi_value js_set_mesh(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *vbo, *idx;
    size_t nv, ni;
    if (!get_typed(env, argv[1], napi_float32_array, &vbo, &nv, true)) return nullptr;
    if (!get_typed(env, argv[2], napi_uint32_array, &idx, &ni, true)) return nullptr;
    if (nv % 9 || ni % 3) { napi_throw_range_error(env, nullptr, "setMesh: vbo needs 9 floats per vertex, idx 3 per triangle"); return nullptr; }
    if (ar_set_mesh(ctx, static_cast<const float *>(vbo), uint32_t(nv / 9), static_cast<const uint32_t *>(idx), uint32_t(ni / 3)))
        return throw_ar(env, ctx, "ar_set_mesh");
    return nullptr;
}

// setMeshIndexed(handle, Float32Array positions (3 per vertex), Uint32Array indices (3 per triangle),
//                Float32Array per-triangle albedo (3 per triangle) | null): ar_set_mesh_indexed
napi_value js_set_mesh_indexed(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *pos, *ind, *col;
    size_t np, ni, nc;
    if (!get_typed(env, argv[1], napi_float32_array, &pos, &np, true)) return nullptr;
    if (!get_typed(env, argv[2], napi_uint32_array, &ind, &ni, true)) return nullptr;
    if (!get_typed(env, argv[3], napi_float32_array, &col, &nc, true)) return nullptr;
    if (np % 3 || ni % 3 || (col && nc != ni)) {
        napi_throw_range_error(env, nullptr, "setMeshIndexed: 3 floats per vertex, 3 indices and (optionally) 3 colour floats per triangle");
        return nullptr;
    }
    if (ar_set_mesh_indexed(ctx, static_cast<const float *>(pos), uint32_t(np / 3), static_cast<const uint32_t *>(ind), uint32_t(ni / 3),
                            static_cast<const float *>(col)))
        return throw_ar(env, ctx, "ar_set_mesh_indexed");
    return nullptr;
}

// readMesh(handle) -> {vbo: Float32Array(9 * V), idx: Uint32Array(3 * T)}: ar_read_mesh
napi_value js_read_mesh(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    uint32_t nv = 0, nt = 0;
    if (ar_read_mesh(ctx, nullptr, nullptr, &nv, &nt)) return throw_ar(env, ctx, "ar_read_mesh");
    void *vdata = nullptr, *idata = nullptr;
    napi_value vbuf, ibuf, vbo, idx, out;
    NAPI_OK(env, napi_create_arraybuffer(env, size_t(nv) * 9 * sizeof(float), &vdata, &vbuf));
    NAPI_OK(env, napi_create_arraybuffer(env, size_t(nt) * 3 * sizeof(uint32_t), &idata, &ibuf));
    if (nt && ar_read_mesh(ctx, static_cast<float *>(vdata), static_cast<uint32_t *>(idata), nullptr, nullptr))
        return throw_ar(env, ctx, "ar_read_mesh");
    NAPI_OK(env, napi_create_typedarray(env, napi_float32_array, size_t(nv) * 9, vbuf, 0, &vbo));
    NAPI_OK(env, napi_create_typedarray(env, napi_uint32_array, size_t(nt) * 3, ibuf, 0, &idx));
    NAPI_OK(env, napi_create_object(env, &out));
    NAPI_OK(env, napi_set_named_property(env, out, "vbo", vbo));
    NAPI_OK(env, napi_set_named_property(env, out, "idx", idx));
    return out;
}

// setMeshUv(handle, Uint16Array(8 * nTris) | null): texel-coordinate rows for ar_set_mesh_uv
napi_value js_set_mesh_uv(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *uv;
    size_t n;
    if (!get_typed(env, argv[1], napi_uint16_array, &uv, &n, true)) return nullptr;
    if (n % 8) { napi_throw_range_error(env, nullptr, "setMeshUv: 8 values per triangle expected"); return nullptr; }
    if (ar_set_mesh_uv(ctx, static_cast<const uint16_t *>(uv), uint32_t(n / 8))) return throw_ar(env, ctx, "ar_set_mesh_uv");
    return nullptr;
}

// setTexture(handle, Uint8Array(width * height * 4) | null, width, height): the atlas for ar_set_texture
napi_value js_set_texture(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *px;
    size_t n;
    if (!get_typed(env, argv[1], napi_uint8_array, &px, &n, true)) return nullptr;
    const int32_t w = get_i32(env, argv[2]), h = get_i32(env, argv[3]);
    if (px && (w < 1 || h < 1 || n < size_t(w) * size_t(h) * 4)) {
        napi_throw_range_error(env, nullptr, "setTexture: pixels shorter than width * height * 4");
        return nullptr;
    }
    if (ar_set_texture(ctx, static_cast<const uint8_t *>(px), w, h)) return throw_ar(env, ctx, "ar_set_texture");
    return nullptr;
}

napi_value js_set_camera(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *p, *v;
    size_t np, nv;
    if (!get_typed(env, argv[1], napi_float32_array, &p, &np, false) || !get_typed(env, argv[2], napi_float32_array, &v, &nv, false)) return nullptr;
    if (np != 16 || nv != 16) { napi_throw_range_error(env, nullptr, "setCamera: two Float32Array(16) expected"); return nullptr; }
    if (ar_set_camera(ctx, static_cast<const float *>(p), static_cast<const float *>(v))) return throw_ar(env, ctx, "ar_set_camera");
    return nullptr;
}

// setModel(handle, Float32Array(16) | null): ar_set_model
napi_value js_set_model(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *m;
    size_t n;
    if (!get_typed(env, argv[1], napi_float32_array, &m, &n, true)) return nullptr;
    if (m && n != 16) { napi_throw_range_error(env, nullptr, "setModel: Float32Array(16) or null expected"); return nullptr; }
    if (ar_set_model(ctx, static_cast<const float *>(m))) return throw_ar(env, ctx, "ar_set_model");
    return nullptr;
}

napi_value js_set_lights(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *a, *d, *c, *pp, *pc;
    size_t na, nd, nc, npp, npc;
    if (!get_typed(env, argv[1], napi_float32_array, &a, &na, false) || !get_typed(env, argv[2], napi_float32_array, &d, &nd, false) ||
        !get_typed(env, argv[3], napi_float32_array, &c, &nc, false) || !get_typed(env, argv[4], napi_float32_array, &pp, &npp, true) ||
        !get_typed(env, argv[5], napi_float32_array, &pc, &npc, true))
        return nullptr;
    if (na < 3 || nd < 3 || nc < 3 || npp != npc || npp % 3) { napi_throw_range_error(env, nullptr, "setLights: bad array sizes"); return nullptr; }
    if (ar_set_lights(ctx, static_cast<const float *>(a), static_cast<const float *>(d), static_cast<const float *>(c), int(npp / 3),
                      static_cast<const float *>(pp), static_cast<const float *>(pc)))
        return throw_ar(env, ctx, "ar_set_lights");
    return nullptr;
}

napi_value js_set_ascii(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *ramp;
    size_t n;
    if (!get_typed(env, argv[1], napi_uint8_array, &ramp, &n, false)) return nullptr;
    bool on = false;
    napi_get_value_bool(env, argv[2], &on);
    if (ar_set_ascii(ctx, static_cast<const uint8_t *>(ramp), int(n), on ? 1 : 0, get_i32(env, argv[3]), get_i32(env, argv[4])))
        return throw_ar(env, ctx, "ar_set_ascii");
    return nullptr;
}

napi_value js_set_strip(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    if (ar_set_strip(ctx, get_i32(env, argv[1]), get_i32(env, argv[2]))) return throw_ar(env, ctx, "ar_set_strip");
    return nullptr;
}

napi_value js_set_option(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    int64_t v = 0;
    napi_get_value_int64(env, argv[2], &v);
    if (ar_set_option(ctx, get_i32(env, argv[1]), v)) return throw_ar(env, ctx, "ar_set_option");
    return nullptr;
}

// render(handle, cols, rows, rgba|null, depth|null, ascii|null) -> true when a frame was drawn,
// false when there is no geometry (buffers untouched), throws on size mismatch or device errors.
napi_value js_render(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *rgba, *depth, *ascii;
    size_t nr, nd, na;
    if (!get_typed(env, argv[3], napi_uint8_array, &rgba, &nr, true) || !get_typed(env, argv[4], napi_float32_array, &depth, &nd, true) ||
        !get_typed(env, argv[5], napi_uint8_array, &ascii, &na, true))
        return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if ((rgba && nr != n * 4) || (depth && nd != n) || (ascii && na != n)) {
        napi_throw_range_error(env, nullptr, "render: output array size does not match cols*rows");
        return nullptr;
    }
    if (ar_render(ctx, cols, rows, static_cast<uint8_t *>(rgba), static_cast<float *>(depth), static_cast<uint8_t *>(ascii)))
        return throw_ar(env, ctx, "ar_render");
    uint32_t c[8];
    napi_value out;
    napi_get_boolean(env, ar_frame_counters(ctx, c) == 0 && c[0] > 0, &out);
    return out;
}

// rtSetScene(handle, packed): the ray backend's scene arrays as packRaytraceScene (js/hip_raytrace_backend.js) lays them
// out - sphereCount / sphereGeom / sphereMaterial, planeCount / planeGeom / planeMaterial, triCount / triA / triB / triC /
// triMaterial, materialCount / materialAlbedo / materialMirror, pointCount / pointPos / pointColor, dirCount / dirDir /
// dirColor, environment - i.e. the uniform arrays of raytrace.js:349-389
napi_value js_rt_set_scene(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    ar_rt_scene sc;
    memset(&sc, 0, sizeof sc);
    bool ok = true;
    auto count = [&](const char *name) {
        napi_value v;
        int32_t x = 0;
        if (napi_get_named_property(env, argv[1], name, &v) != napi_ok || napi_get_value_int32(env, v, &x) != napi_ok) ok = false;
        return int(x);
    };
    auto array = [&](const char *name, napi_typedarray_type type, size_t need) -> const void * {
        napi_value v;
        void *data = nullptr;
        size_t len = 0;
        if (!ok || napi_get_named_property(env, argv[1], name, &v) != napi_ok) { ok = false; return nullptr; }
        if (!get_typed(env, v, type, &data, &len, need == 0)) { ok = false; return nullptr; }
        if (len < need) { ok = false; napi_throw_range_error(env, nullptr, "rtSetScene: array shorter than its count"); return nullptr; }
        return need ? data : nullptr;
    };
    sc.n_spheres = count("sphereCount"); sc.n_planes = count("planeCount"); sc.n_tris = count("triCount");
    sc.n_mats = count("materialCount"); sc.n_pl = count("pointCount"); sc.n_dl = count("dirCount");
    if (!ok || sc.n_spheres < 0 || sc.n_planes < 0 || sc.n_tris < 0 || sc.n_mats < 0 || sc.n_pl < 0 || sc.n_dl < 0) {
        napi_throw_type_error(env, nullptr, "rtSetScene: counts missing or negative");
        return nullptr;
    }
    sc.spheres4 = static_cast<const float *>(array("sphereGeom", napi_float32_array, size_t(sc.n_spheres) * 4));
    sc.sphere_mat = static_cast<const int32_t *>(array("sphereMaterial", napi_int32_array, size_t(sc.n_spheres)));
    sc.planes4 = static_cast<const float *>(array("planeGeom", napi_float32_array, size_t(sc.n_planes) * 4));
    sc.plane_mat = static_cast<const int32_t *>(array("planeMaterial", napi_int32_array, size_t(sc.n_planes)));
    sc.tri_a = static_cast<const float *>(array("triA", napi_float32_array, size_t(sc.n_tris) * 3));
    sc.tri_b = static_cast<const float *>(array("triB", napi_float32_array, size_t(sc.n_tris) * 3));
    sc.tri_c = static_cast<const float *>(array("triC", napi_float32_array, size_t(sc.n_tris) * 3));
    sc.tri_mat = static_cast<const int32_t *>(array("triMaterial", napi_int32_array, size_t(sc.n_tris)));
    sc.mat_albedo = static_cast<const float *>(array("materialAlbedo", napi_float32_array, size_t(sc.n_mats) * 3));
    sc.mat_reflective = static_cast<const int32_t *>(array("materialMirror", napi_int32_array, size_t(sc.n_mats)));
    sc.pl_pos = static_cast<const float *>(array("pointPos", napi_float32_array, size_t(sc.n_pl) * 3));
    sc.pl_col = static_cast<const float *>(array("pointColor", napi_float32_array, size_t(sc.n_pl) * 3));
    sc.dl_dir = static_cast<const float *>(array("dirDir", napi_float32_array, size_t(sc.n_dl) * 3));
    sc.dl_col = static_cast<const float *>(array("dirColor", napi_float32_array, size_t(sc.n_dl) * 3));
    const float *env3 = static_cast<const float *>(array("environment", napi_float32_array, 3));
    if (!ok || !env3) {
        bool pending = false;
        napi_is_exception_pending(env, &pending);
        if (!pending) napi_throw_type_error(env, nullptr, "rtSetScene: scene arrays missing or of the wrong type");
        return nullptr;
    }
    for (int k = 0; k < 3; k++) sc.env[k] = env3[k];
    if (ar_rt_set_scene(ctx, &sc)) return throw_ar(env, ctx, "ar_rt_set_scene");
    return nullptr;
}

// rtRender(handle, cols, rows, Float32Array camPos(3), yaw, pitch, fovY, pixAspect, rgba|null, ascii|null) -> true when a scene was drawn
napi_value js_rt_render(napi_env env, napi_callback_info info) {
    napi_value argv[10];
    if (!get_args(env, info, 10, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *cam, *rgba, *ascii;
    size_t nc, nr, na;
    if (!get_typed(env, argv[3], napi_float32_array, &cam, &nc, false) || !get_typed(env, argv[8], napi_uint8_array, &rgba, &nr, true) ||
        !get_typed(env, argv[9], napi_uint8_array, &ascii, &na, true))
        return nullptr;
    double f[4];
    for (int k = 0; k < 4; k++)
        if (napi_get_value_double(env, argv[4 + k], &f[k]) != napi_ok) { napi_throw_type_error(env, nullptr, "rtRender: numbers expected"); return nullptr; }
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if (nc != 3 || (rgba && nr != n * 4) || (ascii && na != n)) {
        napi_throw_range_error(env, nullptr, "rtRender: array size does not match (camera 3, rgba cols*rows*4, ascii cols*rows)");
        return nullptr;
    }
    if (ar_rt_render(ctx, cols, rows, static_cast<const float *>(cam), float(f[0]), float(f[1]), float(f[2]), float(f[3]),
                     static_cast<uint8_t *>(rgba), static_cast<uint8_t *>(ascii)))
        return throw_ar(env, ctx, "ar_rt_render");
    return nullptr;
}

napi_value js_quantize(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *rgba, *ascii;
    size_t nr, na;
    const int cols = get_i32(env, argv[2]), rows = get_i32(env, argv[3]);
    if (!get_typed(env, argv[1], napi_uint8_array, &rgba, &nr, false) || !get_typed(env, argv[4], napi_uint8_array, &ascii, &na, false)) return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if (nr != n * 4 || na != n) { napi_throw_range_error(env, nullptr, "quantize: array sizes do not match cols*rows"); return nullptr; }
    if (ar_quantize(ctx, static_cast<const uint8_t *>(rgba), cols, rows, static_cast<uint8_t *>(ascii))) return throw_ar(env, ctx, "ar_quantize");
    return nullptr;
}

// quantizeStatic(handle, rgba, cols, rows, outAscii): ar_quantize_static (no device allocation on the way)
napi_value js_quantize_static(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *rgba, *ascii;
    size_t nr, na;
    const int cols = get_i32(env, argv[2]), rows = get_i32(env, argv[3]);
    if (!get_typed(env, argv[1], napi_uint8_array, &rgba, &nr, false) || !get_typed(env, argv[4], napi_uint8_array, &ascii, &na, false)) return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if (nr != n * 4 || na != n) { napi_throw_range_error(env, nullptr, "quantizeStatic: array sizes do not match cols*rows"); return nullptr; }
    if (ar_quantize_static(ctx, static_cast<const uint8_t *>(rgba), cols, rows, static_cast<uint8_t *>(ascii))) return throw_ar(env, ctx, "ar_quantize_static");
    return nullptr;
}

// quantizeColors(handle, rgba, cols, rows, steps, grayscale, outUint32Array)
napi_value js_quantize_colors(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *rgba, *colors;
    size_t nr, nc;
    const int cols = get_i32(env, argv[2]), rows = get_i32(env, argv[3]);
    if (!get_typed(env, argv[1], napi_uint8_array, &rgba, &nr, false) || !get_typed(env, argv[6], napi_uint32_array, &colors, &nc, false)) return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if (nr != n * 4 || nc != n) { napi_throw_range_error(env, nullptr, "quantizeColors: array sizes do not match cols*rows"); return nullptr; }
    bool gray = false;
    napi_get_value_bool(env, argv[5], &gray);
    if (ar_quantize_colors(ctx, static_cast<const uint8_t *>(rgba), cols, rows, get_i32(env, argv[4]), gray ? 1 : 0, static_cast<uint32_t *>(colors)))
        return throw_ar(env, ctx, "ar_quantize_colors");
    return nullptr;
}

// present(handle, rgba, cols, rows, atlasAlpha, cellW, cellH, gamma, grayscale, transparentBg, outImage)
napi_value js_present(napi_env env, napi_callback_info info) {
    napi_value argv[11];
    if (!get_args(env, info, 11, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *rgba, *atlas, *image;
    size_t nr, na, ni;
    const int cols = get_i32(env, argv[2]), rows = get_i32(env, argv[3]), cw = get_i32(env, argv[5]), ch = get_i32(env, argv[6]);
    if (!get_typed(env, argv[1], napi_uint8_array, &rgba, &nr, false) || !get_typed(env, argv[4], napi_uint8_array, &atlas, &na, false) ||
        !get_typed(env, argv[10], napi_uint8_array, &image, &ni, false))
        return nullptr;
    if (cols <= 0 || rows <= 0 || cw <= 0 || ch <= 0) { napi_throw_range_error(env, nullptr, "present: bad grid or cell size"); return nullptr; }
    const size_t n = size_t(cols) * size_t(rows);
    if (nr != n * 4 || na != size_t(256) * size_t(cw + 4) * size_t(ch + 4) || ni != n * size_t(cw) * size_t(ch) * 4) {
        napi_throw_range_error(env, nullptr, "present: array sizes do not match the grid / atlas layout");
        return nullptr;
    }
    double gamma = 1.0;
    bool gray = false, transparent = false;
    napi_get_value_double(env, argv[7], &gamma);
    napi_get_value_bool(env, argv[8], &gray);
    napi_get_value_bool(env, argv[9], &transparent);
    if (ar_present(ctx, static_cast<const uint8_t *>(rgba), cols, rows, static_cast<const uint8_t *>(atlas), cw, ch, gamma, gray ? 1 : 0,
                   transparent ? 1 : 0, static_cast<uint8_t *>(image)))
        return throw_ar(env, ctx, "ar_present");
    return nullptr;
}

void finalize_host(napi_env, void *data, void *) { ar_host_free(data); }

// allocHost(handle, bytes) -> ArrayBuffer over page-locked host memory (freed when the buffer is collected)
napi_value js_alloc_host(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    double bytes = 0;
    if (napi_get_value_double(env, argv[1], &bytes) != napi_ok || !(bytes >= 0) || bytes > 1e12) {
        napi_throw_range_error(env, nullptr, "allocHost: bad size");
        return nullptr;
    }
    void *p = ar_host_alloc(ctx, size_t(bytes));
    if (!p) return throw_ar(env, ctx, "ar_host_alloc");
    memset(p, 0, size_t(bytes));
    napi_value buf;
    if (napi_create_external_arraybuffer(env, p, size_t(bytes), finalize_host, nullptr, &buf) != napi_ok) {
        ar_host_free(p);
        napi_throw_error(env, nullptr, "allocHost: cannot wrap the buffer");
        return nullptr;
    }
    return buf;
}

napi_value js_frame_counters(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    uint32_t c[8];
    if (ar_frame_counters(ctx, c)) return throw_ar(env, ctx, "ar_frame_counters");
    napi_value arr;
    NAPI_OK(env, napi_create_array_with_length(env, 8, &arr));
    for (uint32_t k = 0; k < 8; k++) {
        napi_value v;
        napi_create_uint32(env, c[k], &v);
        napi_set_element(env, arr, k, v);
    }
    return arr;
}

// A device or stream address crosses the boundary as a BigInt (exact) or, from callers that computed it, a Number.
bool get_address(napi_env env, napi_value v, void **out) {
    napi_valuetype vt;
    napi_typeof(env, v, &vt);
    *out = nullptr;
    if (vt == napi_null || vt == napi_undefined) return true;
    if (vt == napi_bigint) {
        uint64_t x = 0;
        bool lossless = false;
        if (napi_get_value_bigint_uint64(env, v, &x, &lossless) != napi_ok || !lossless) {
            napi_throw_range_error(env, nullptr, "ascii_rast addon: address out of range");
            return false;
        }
        *out = reinterpret_cast<void *>(uintptr_t(x));
        return true;
    }
    if (vt == napi_number) {
        double d = 0;
        napi_get_value_double(env, v, &d);
        if (!(d >= 0) || d > 9007199254740992.0) { napi_throw_range_error(env, nullptr, "ascii_rast addon: address out of range"); return false; }
        *out = reinterpret_cast<void *>(uintptr_t(d));
        return true;
    }
    napi_throw_type_error(env, nullptr, "ascii_rast addon: address (BigInt, Number or null) expected");
    return false;
}

napi_value make_address(napi_env env, const void *p) {
    napi_value v = nullptr;
    if (napi_create_bigint_uint64(env, uint64_t(reinterpret_cast<uintptr_t>(p)), &v) != napi_ok) {
        napi_throw_error(env, nullptr, "ascii_rast addon: cannot create a BigInt");
        return nullptr;
    }
    return v;
}

bool get_ticket(napi_env env, napi_value v, uint64_t *ticket) {
    double d = -1;
    if (napi_get_value_double(env, v, &d) != napi_ok || !(d >= 0) || d > 9007199254740992.0) {
        napi_throw_type_error(env, nullptr, "ascii_rast addon: ticket (a Number returned by readAsync) expected");
        return false;
    }
    *ticket = uint64_t(d);
    return true;
}

// renderAsync(handle, cols, rows): ar_render_async - enqueue a frame and return
napi_value js_render_async(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    if (ar_render_async(ctx, get_i32(env, argv[1]), get_i32(env, argv[2]))) return throw_ar(env, ctx, "ar_render_async");
    return nullptr;
}

// readAsync(handle, cols, rows, rgba|null, depth|null, ascii|null, ids|null) -> ticket (Number): ar_read_async.  The
// arrays must stay alive (and should be page-locked, allocHost) until wait(ticket) has returned.
napi_value js_read_async(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *rgba, *depth, *ascii, *ids;
    size_t nr, nd, na, ni;
    if (!get_typed(env, argv[3], napi_uint8_array, &rgba, &nr, true) || !get_typed(env, argv[4], napi_float32_array, &depth, &nd, true) ||
        !get_typed(env, argv[5], napi_uint8_array, &ascii, &na, true) || !get_typed(env, argv[6], napi_uint32_array, &ids, &ni, true))
        return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if ((rgba && nr != n * 4) || (depth && nd != n) || (ascii && na != n) || (ids && ni != n)) {
        napi_throw_range_error(env, nullptr, "readAsync: output array size does not match cols*rows");
        return nullptr;
    }
    uint64_t ticket = 0;
    if (ar_read_async(ctx, static_cast<uint8_t *>(rgba), static_cast<float *>(depth), static_cast<uint8_t *>(ascii), static_cast<uint32_t *>(ids), &ticket))
        return throw_ar(env, ctx, "ar_read_async");
    napi_value out;
    NAPI_OK(env, napi_create_double(env, double(ticket), &out));
    return out;
}

napi_value js_wait(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    uint64_t ticket;
    if (!ctx || !get_ticket(env, argv[1], &ticket)) return nullptr;
    if (ar_wait(ctx, ticket)) return throw_ar(env, ctx, "ar_wait");
    return nullptr;
}

// poll(handle, ticket) -> true when the frame and its copies are complete
napi_value js_poll(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    uint64_t ticket;
    if (!ctx || !get_ticket(env, argv[1], &ticket)) return nullptr;
    const int rc = ar_poll(ctx, ticket);
    if (rc < 0) return throw_ar(env, ctx, "ar_poll");
    napi_value out;
    napi_get_boolean(env, rc == 1, &out);
    return out;
}

napi_value js_sync(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    if (ar_sync(ctx)) return throw_ar(env, ctx, "ar_sync");
    return nullptr;
}

// read(handle, cols, rows, rgba|null, depth|null, ascii|null, ids|null): ar_read (blocking copy of the last frame)
napi_value js_read(napi_env env, napi_callback_info info) {
    napi_value argv[7];
    if (!get_args(env, info, 7, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *rgba, *depth, *ascii, *ids;
    size_t nr, nd, na, ni;
    if (!get_typed(env, argv[3], napi_uint8_array, &rgba, &nr, true) || !get_typed(env, argv[4], napi_float32_array, &depth, &nd, true) ||
        !get_typed(env, argv[5], napi_uint8_array, &ascii, &na, true) || !get_typed(env, argv[6], napi_uint32_array, &ids, &ni, true))
        return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if ((rgba && nr != n * 4) || (depth && nd != n) || (ascii && na != n) || (ids && ni != n)) {
        napi_throw_range_error(env, nullptr, "read: output array size does not match cols*rows");
        return nullptr;
    }
    if (ar_read(ctx, static_cast<uint8_t *>(rgba), static_cast<float *>(depth), static_cast<uint8_t *>(ascii), static_cast<uint32_t *>(ids)))
        return throw_ar(env, ctx, "ar_read");
    return nullptr;
}

// deviceBuffer(handle, which) -> BigInt device address (0n before the first frame): ar_device_buffer
napi_value js_device_buffer(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    return make_address(env, ar_device_buffer(ctx, get_i32(env, argv[1])));
}

// stream(handle) -> BigInt hipStream_t: ar_stream
napi_value js_stream(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    return make_address(env, ar_stream(ctx));
}

napi_value js_set_stream(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    void *s;
    if (!ctx || !get_address(env, argv[1], &s)) return nullptr;
    if (ar_set_stream(ctx, s)) return throw_ar(env, ctx, "ar_set_stream");
    return nullptr;
}

// setAsciiTarget(handle, address|null, firstRow): ar_set_ascii_target
napi_value js_set_ascii_target(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    void *p;
    if (!ctx || !get_address(env, argv[1], &p)) return nullptr;
    if (ar_set_ascii_target(ctx, p, get_i32(env, argv[2]))) return throw_ar(env, ctx, "ar_set_ascii_target");
    return nullptr;
}

napi_value js_stream_wait_frame(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    void *s;
    if (!ctx || !get_address(env, argv[1], &s)) return nullptr;
    if (ar_stream_wait_frame(ctx, s)) return throw_ar(env, ctx, "ar_stream_wait_frame");
    return nullptr;
}

napi_value js_frame_wait_stream(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    void *s;
    if (!ctx || !get_address(env, argv[1], &s)) return nullptr;
    if (ar_frame_wait_stream(ctx, s)) return throw_ar(env, ctx, "ar_frame_wait_stream");
    return nullptr;
}

napi_value js_set_profiling(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    bool on = false;
    napi_get_value_bool(env, argv[1], &on);
    if (ar_set_profiling(ctx, on ? 1 : 0)) return throw_ar(env, ctx, "ar_set_profiling");
    return nullptr;
}

// kernelStats(handle) -> { kernelName: { launches, ms } } for the kernels launched since the last call: ar_kernel_stats
napi_value js_kernel_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const char *names[AR_N_KERNELS];
    uint32_t launches[AR_N_KERNELS];
    float ms[AR_N_KERNELS];
    if (ar_kernel_stats(ctx, names, launches, ms)) return throw_ar(env, ctx, "ar_kernel_stats");
    napi_value out;
    NAPI_OK(env, napi_create_object(env, &out));
    for (int k = 0; k < AR_N_KERNELS; k++) {
        if (!names[k] || !names[k][0] || !launches[k]) continue;
        napi_value rec, n, t;
        NAPI_OK(env, napi_create_object(env, &rec));
        NAPI_OK(env, napi_create_uint32(env, launches[k], &n));
        NAPI_OK(env, napi_create_double(env, double(ms[k]), &t));
        NAPI_OK(env, napi_set_named_property(env, rec, "launches", n));
        NAPI_OK(env, napi_set_named_property(env, rec, "ms", t));
        NAPI_OK(env, napi_set_named_property(env, out, names[k], rec));
    }
    return out;
}

// readColors(handle, cols, rows, steps, grayscale, outUint32Array): ar_read_colors of the last rendered frame
napi_value js_read_colors(napi_env env, napi_callback_info info) {
    napi_value argv[6];
    if (!get_args(env, info, 6, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *colors;
    size_t nc;
    if (!get_typed(env, argv[5], napi_uint32_array, &colors, &nc, false)) return nullptr;
    if (nc != size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0)) { napi_throw_range_error(env, nullptr, "readColors: array size does not match cols*rows"); return nullptr; }
    bool gray = false;
    napi_get_value_bool(env, argv[4], &gray);
    if (ar_read_colors(ctx, get_i32(env, argv[3]), gray ? 1 : 0, static_cast<uint32_t *>(colors))) return throw_ar(env, ctx, "ar_read_colors");
    return nullptr;
}

napi_value js_set_id_output(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    bool on = false;
    napi_get_value_bool(env, argv[1], &on);
    if (ar_set_id_output(ctx, on ? 1 : 0)) return throw_ar(env, ctx, "ar_set_id_output");
    return nullptr;
}

// renderBatchAsync(handle, cols, rows, Float32Array cameras (32 floats per frame: projection, view), asciiStride): ar_render_batch_async
napi_value js_render_batch_async(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *cams;
    size_t n;
    if (!get_typed(env, argv[3], napi_float32_array, &cams, &n, false)) return nullptr;
    if (n == 0 || n % 32) { napi_throw_range_error(env, nullptr, "renderBatchAsync: 32 floats (projection, view) per frame expected"); return nullptr; }
    double stride = 0;
    napi_get_value_double(env, argv[4], &stride);
    if (!(stride >= 0)) stride = 0;
    if (ar_render_batch_async(ctx, get_i32(env, argv[1]), get_i32(env, argv[2]), int(n / 32), static_cast<const float *>(cams), size_t(stride)))
        return throw_ar(env, ctx, "ar_render_batch_async");
    return nullptr;
}

// readBatch(handle, cols, rows, frame, rgba|null, depth|null, ascii|null, ids|null): ar_read_batch
napi_value js_read_batch(napi_env env, napi_callback_info info) {
    napi_value argv[8];
    if (!get_args(env, info, 8, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    const int cols = get_i32(env, argv[1]), rows = get_i32(env, argv[2]);
    void *rgba, *depth, *ascii, *ids;
    size_t nr, nd, na, ni;
    if (!get_typed(env, argv[4], napi_uint8_array, &rgba, &nr, true) || !get_typed(env, argv[5], napi_float32_array, &depth, &nd, true) ||
        !get_typed(env, argv[6], napi_uint8_array, &ascii, &na, true) || !get_typed(env, argv[7], napi_uint32_array, &ids, &ni, true))
        return nullptr;
    const size_t n = size_t(cols > 0 ? cols : 0) * size_t(rows > 0 ? rows : 0);
    if ((rgba && nr != n * 4) || (depth && nd != n) || (ascii && na != n) || (ids && ni != n)) {
        napi_throw_range_error(env, nullptr, "readBatch: output array size does not match cols*rows");
        return nullptr;
    }
    if (ar_read_batch(ctx, get_i32(env, argv[3]), static_cast<uint8_t *>(rgba), static_cast<float *>(depth), static_cast<uint8_t *>(ascii), static_cast<uint32_t *>(ids)))
        return throw_ar(env, ctx, "ar_read_batch");
    return nullptr;
}

// updatePositions(handle, Float32Array data, floatsPerVertex (3 | 9), flags): ar_update_positions
napi_value js_update_positions(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    ar_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return nullptr;
    void *data;
    size_t n;
    if (!get_typed(env, argv[1], napi_float32_array, &data, &n, false)) return nullptr;
    const int fpv = get_i32(env, argv[2]);
    if ((fpv != 3 && fpv != 9) || n % size_t(fpv)) { napi_throw_range_error(env, nullptr, "updatePositions: 3 or 9 floats per vertex"); return nullptr; }
    if (ar_update_positions(ctx, static_cast<const float *>(data), uint32_t(n / size_t(fpv)), fpv, get_i32(env, argv[3])))
        return throw_ar(env, ctx, "ar_update_positions");
    return nullptr;
}

ar_group *get_group(napi_env env, napi_value v) {
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !static_cast<GroupHandle *>(p)->group) {
        napi_throw_type_error(env, nullptr, "ascii_rast addon: invalid or destroyed group handle");
        return nullptr;
    }
    return static_cast<GroupHandle *>(p)->group;
}

napi_value throw_group(napi_env env, ar_group *g, const char *what) {
    char msg[384];
    snprintf(msg, sizeof msg, "%s: %s", what, ar_group_last_error(g));
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

// stripRows(rows, nStrips, strip) -> [row0, row1]: ar_strip_rows
napi_value js_strip_rows(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    int row0 = 0, row1 = 0;
    if (ar_strip_rows(get_i32(env, argv[0]), get_i32(env, argv[1]), get_i32(env, argv[2]), &row0, &row1)) {
        napi_throw_range_error(env, nullptr, "stripRows: bad arguments");
        return nullptr;
    }
    napi_value arr, a, b;
    NAPI_OK(env, napi_create_array_with_length(env, 2, &arr));
    napi_create_int32(env, row0, &a);
    napi_create_int32(env, row1, &b);
    napi_set_element(env, arr, 0, a);
    napi_set_element(env, arr, 1, b);
    return arr;
}

// groupCreate(Int32Array deviceIds, batch) -> group handle: ar_group_create
napi_value js_group_create(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    void *ids;
    size_t n;
    if (!get_typed(env, argv[0], napi_int32_array, &ids, &n, false)) return nullptr;
    ar_group *g = ar_group_create(int(n), static_cast<const int *>(ids), get_i32(env, argv[1]));
    if (!g) return throw_group(env, nullptr, "ar_group_create");
    GroupHandle *h = new GroupHandle{g};
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, h, finalize_group, nullptr, &ext));
    return ext;
}

napi_value js_group_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    void *p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        GroupHandle *h = static_cast<GroupHandle *>(p);
        if (h->group) { ar_group_destroy(h->group); h->group = nullptr; }
    }
    return nullptr;
}

// groupCtx(group, member) -> renderer handle of that member (owned by the group; valid until groupDestroy): ar_group_ctx
napi_value js_group_ctx(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    ar_group *g = get_group(env, argv[0]);
    if (!g) return nullptr;
    ar_ctx *ctx = ar_group_ctx(g, get_i32(env, argv[1]));
    if (!ctx) { napi_throw_range_error(env, nullptr, "groupCtx: no such member"); return nullptr; }
    Handle *h = new Handle{ctx, false};
    napi_value ext;
    NAPI_OK(env, napi_create_external(env, h, finalize_handle, nullptr, &ext));
    return ext;
}

// groupRenderAsync(group, cols, rows, proj, view): ar_group_render_async
napi_value js_group_render_async(napi_env env, napi_callback_info info) {
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    ar_group *g = get_group(env, argv[0]);
    if (!g) return nullptr;
    void *p, *v;
    size_t np, nv;
    if (!get_typed(env, argv[3], napi_float32_array, &p, &np, false) || !get_typed(env, argv[4], napi_float32_array, &v, &nv, false)) return nullptr;
    if (np != 16 || nv != 16) { napi_throw_range_error(env, nullptr, "groupRenderAsync: two Float32Array(16) expected"); return nullptr; }
    if (ar_group_render_async(g, get_i32(env, argv[1]), get_i32(env, argv[2]), static_cast<const float *>(p), static_cast<const float *>(v)))
        return throw_group(env, g, "ar_group_render_async");
    return nullptr;
}

napi_value js_group_sync(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_group *g = get_group(env, argv[0]);
    if (!g) return nullptr;
    if (ar_group_sync(g)) return throw_group(env, g, "ar_group_sync");
    return nullptr;
}

napi_value js_group_frames_gathered(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    ar_group *g = get_group(env, argv[0]);
    if (!g) return nullptr;
    napi_value out;
    napi_create_int32(env, ar_group_frames_gathered(g), &out);
    return out;
}

// groupRead(group, member, frame, outAscii Uint8Array(cols*rows)): ar_group_read (the caller sizes the array to the last grid)
napi_value js_group_read(napi_env env, napi_callback_info info) {
    napi_value argv[4];
    if (!get_args(env, info, 4, argv)) return nullptr;
    ar_group *g = get_group(env, argv[0]);
    if (!g) return nullptr;
    void *out;
    size_t n;
    if (!get_typed(env, argv[3], napi_uint8_array, &out, &n, false)) return nullptr;
    if (ar_group_read(g, get_i32(env, argv[1]), get_i32(env, argv[2]), static_cast<uint8_t *>(out))) return throw_group(env, g, "ar_group_read");
    return nullptr;
}

napi_value init(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        {"create", nullptr, js_create, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"destroy", nullptr, js_destroy, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setMesh", nullptr, js_set_mesh, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setMeshIndexed", nullptr, js_set_mesh_indexed, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readMesh", nullptr, js_read_mesh, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setMeshUv", nullptr, js_set_mesh_uv, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setTexture", nullptr, js_set_texture, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setCamera", nullptr, js_set_camera, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setModel", nullptr, js_set_model, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setLights", nullptr, js_set_lights, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setAscii", nullptr, js_set_ascii, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setStrip", nullptr, js_set_strip, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setOption", nullptr, js_set_option, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"render", nullptr, js_render, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"rtSetScene", nullptr, js_rt_set_scene, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"rtRender", nullptr, js_rt_render, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"quantize", nullptr, js_quantize, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"quantizeStatic", nullptr, js_quantize_static, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"frameCounters", nullptr, js_frame_counters, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"allocHost", nullptr, js_alloc_host, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"quantizeColors", nullptr, js_quantize_colors, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"present", nullptr, js_present, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"renderBatchAsync", nullptr, js_render_batch_async, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readBatch", nullptr, js_read_batch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"updatePositions", nullptr, js_update_positions, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"stripRows", nullptr, js_strip_rows, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupCreate", nullptr, js_group_create, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupDestroy", nullptr, js_group_destroy, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupCtx", nullptr, js_group_ctx, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupRenderAsync", nullptr, js_group_render_async, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupSync", nullptr, js_group_sync, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupFramesGathered", nullptr, js_group_frames_gathered, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupRead", nullptr, js_group_read, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"renderAsync", nullptr, js_render_async, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readAsync", nullptr, js_read_async, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"wait", nullptr, js_wait, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"poll", nullptr, js_poll, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"sync", nullptr, js_sync, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"read", nullptr, js_read, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"deviceBuffer", nullptr, js_device_buffer, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"stream", nullptr, js_stream, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setStream", nullptr, js_set_stream, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setAsciiTarget", nullptr, js_set_ascii_target, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"streamWaitFrame", nullptr, js_stream_wait_frame, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"frameWaitStream", nullptr, js_frame_wait_stream, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setProfiling", nullptr, js_set_profiling, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"kernelStats", nullptr, js_kernel_stats, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readColors", nullptr, js_read_colors, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setIdOutput", nullptr, js_set_id_output, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
