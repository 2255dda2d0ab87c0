// ascii_rast.hip - MI355X (gfx950) 3D -> ASCII rasterizer: HIP kernels and the C ABI around them.
//
// Two resolve paths produce identical frames (nothing leaves HBM between their stages):
//   tiled (default)   k_setup -> k_binscan -> k_fill -> k_tile -> smoothing (k_ascii_counts for ramps of at most ten
//                     glyphs and radius <= 2, else k_ascii_plane; k_ascii when the frame carries ASCII overrides)
//                     triangles are culled and binned to screen bins, then resolved per 64x32-cell
//                     tile in LDS; described in front of k_tristream.
//   direct            k_raster -> k_raster_medium -> k_raster_large -> k_shade -> smoothing (as above)
//                     k_raster: one lane per triangle (vertex transform, near clip, edge setup, cull,
//                     cell range); small triangles are scanned in place, larger ones queued as setup
//                     records for k_raster_medium (one wavefront per record, 8x8 cells per step) and
//                     k_raster_large (one lane per cell scanning the short list of screen-filling
//                     records); k_shade: one lane per cell (winner's varyings, Lambert + point lights,
//                     RGBA8, depth, ramp quantisation; ASCII written here when smoothing is off).
//                     Used for grids beyond the tiled path's tile budget, or on request.
// Visibility is resolved with one 64-bit key per cell, (depth bits << 32) | order, reduced with unsigned
// min: smaller depth wins, equal depth goes to the earlier triangle (order = 2*triangle + sub), and the result
// does not depend on the order in which lanes, waves or kernels reach a cell.
//
// Arithmetic follows a fixed contract (binary32 transforms and shading, binary64 edge functions,
// no fused multiply-add, correctly rounded division and square root) so that the output is
// reproducible bit for bit; build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ascii_rast.h"

#include "raster_common.h"
#include "shading.h"
#include "raster_direct.h"
#include "tiled_binning.h"
#include "tiled_resolve.h"
#include "ascii_stage.h"
#include "raytrace.h"
#include "context.h"

// ---- ar_quantize without allocations ------------------------------------------------------------------------
// ar_quantize stages the caller's RGBA grid and the ASCII plane in device memory that it allocates (and grows) on
// demand.  The variant below stages them in a fixed array that is part of the loaded code object instead, so that
// it never calls hipMalloc / hipFree: for callers that cannot allocate (serving loops and harnesses that forbid it, code
// that runs while another thread captures a HIP graph; the call itself copies from and to pageable host memory and
// waits for its stream, so it is not itself capturable).  Every GPU the library runs on holds its own copy of the array until the process
// ends; all calls on a GPU share it, so calls that overlap on one GPU race, and it keeps the last call's contents
// (nothing in it has to start at zero: a call overwrites every byte it later reads).
static __device__ __align__(256) unsigned char scratch[size_t(AR_STATIC_MAX_CELLS) * 6];  // RGBA staging, the ASCII plane, the ramp-index plane

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU (the symbol's name is only a host-side handle)

// Address of this device's copy of the array; looked up once per device, with that device current.
int static_scratch(ar_ctx *ctx, unsigned char **out) {
    if (ctx->device < 0 || ctx->device >= kMaxDevices) return fail(ctx, AR_E_ARG, "device %d beyond the %d the static staging array is tracked for", ctx->device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[ctx->device]) {
        AR_HIP(ctx, hipSetDevice(ctx->device));
        void *p = nullptr;
        AR_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(ctx, AR_E_HIP, "hipGetSymbolAddress returned a null address for the static staging array");
        g_scratch_ptr[ctx->device] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[ctx->device];
    return 0;
}

}  // namespace

extern "C" {

ar_ctx *ar_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(nullptr, AR_E_HIP, "no HIP device available (%s)", hipGetErrorString(e)); return nullptr; }
    if (device_id < 0 || device_id >= n) { fail(nullptr, AR_E_ARG, "device %d out of range (0..%d)", device_id, n - 1); return nullptr; }
    ar_ctx *ctx = new (std::nothrow) ar_ctx();
    if (!ctx) { fail(nullptr, AR_E_NOMEM, "out of host memory"); return nullptr; }
    ctx->device = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess &&
              (ctx->stream = ctx->own_stream) != nullptr && hipEventCreateWithFlags(&ctx->ev_ext, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&ctx->d_medium, size_t(kMediumCapacity) * sizeof(TriRec)) == hipSuccess &&
              hipMalloc(&ctx->d_large, size_t(kLargeCapacity) * sizeof(TriRec)) == hipSuccess &&
              hipMalloc(&ctx->d_counters, 4 * sizeof(uint32_t)) == hipSuccess &&
              hipMemset(ctx->d_counters, 0, 4 * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc(&ctx->d_state, kSets * sizeof(BinState)) == hipSuccess &&
              hipMemset(ctx->d_state, 0, kSets * sizeof(BinState)) == hipSuccess;
    ctx->fstream[0] = ctx->own_stream;
    for (int k = 1; ok && k < kSets; k++) ok = hipStreamCreateWithFlags(&ctx->fstream[k], hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k <= kSets; k++) ok = hipEventCreateWithFlags(&ctx->ev_out[k], hipEventDisableTiming) == hipSuccess;
    for (int k = 0; ok && k < kSets; k++) ok = hipEventCreateWithFlags(&ctx->ev_upd[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->ev_staged, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ctx->ev_taken, hipEventDisableTiming) == hipSuccess &&
         hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k <= kSets; k++)
        for (int j = 0; ok && j < kTicketRing; j++) ok = hipEventCreateWithFlags(&ctx->ev_done[k][j], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        fail(nullptr, AR_E_HIP, "context setup failed: %s", hipGetErrorString(hipGetLastError()));
        ar_destroy(ctx);
        return nullptr;
    }
    {   // this device's copy of the static staging array (ar_quantize_static): its address is looked up here, once per
        // device; a failure is that entry point's to report (it looks again), never the context's
        unsigned char *p = nullptr;
        (void)static_scratch(ctx, &p);
        ctx->err[0] = '\0';
    }
    memset(&ctx->fp, 0, sizeof ctx->fp);
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(ctx->proj, ident, sizeof ident);
    memcpy(ctx->view, ident, sizeof ident);
    // defaults of the reference: raster.js:66-68, config.js:15,26-34 with ascii_pass.js:149-152
    const float amb[3] = {0.15f, 0.18f, 0.22f}, dir[3] = {0.25f, -1.0f, 0.15f}, col[3] = {1.2f, 1.15f, 1.1f};
    ar_set_lights(ctx, amb, dir, col, 0, nullptr, nullptr);
    ar_set_ascii(ctx, reinterpret_cast<const uint8_t *>("@%#*+=-:. "), 10, 1, 2, 12);
    return ctx;
}

void ar_destroy(ar_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int k = 0; k < kSets; k++)
        if (ctx->fstream[k]) (void)hipStreamSynchronize(ctx->fstream[k]);
    if (ctx->stream && ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->stamps) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    free_targets(ctx);
    for (int k = 0; k < kSets; k++) free_batch_set(ctx->batch[k]);
    (void)hipFree(ctx->d_vbo); (void)hipFree(ctx->d_idx);
    (void)hipFree(ctx->d_medium); (void)hipFree(ctx->d_large); (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii); (void)hipFree(ctx->d_q_plane); (void)hipFree(ctx->d_q_colors);
    (void)hipFree(ctx->d_p_atlas); (void)hipFree(ctx->d_p_cov); (void)hipFree(ctx->d_p_image);
    (void)hipFree(ctx->d_state); (void)hipFree(ctx->d_tripos); (void)hipFree(ctx->d_triattr); (void)hipFree(ctx->d_tri);
    (void)hipFree(ctx->d_triuv); (void)hipFree(ctx->d_atlas); (void)hipFree(ctx->d_rt);
    (void)hipFree(ctx->d_tri_t); (void)hipFree(ctx->d_perm); (void)hipFree(ctx->d_box);
    for (int k = 0; k < kSets; k++) {
        (void)hipFree(ctx->d_entries[k]); (void)hipFree(ctx->d_surv[k]); (void)hipFree(ctx->d_sid[k]); (void)hipFree(ctx->d_surv_count[k]);
        (void)hipFree(ctx->d_dircount[k]);
    }
    for (int k = 0; k <= kSets; k++) {
        if (ctx->ev_out[k]) (void)hipEventDestroy(ctx->ev_out[k]);
        for (int j = 0; j < kTicketRing; j++)
            if (ctx->ev_done[k][j]) (void)hipEventDestroy(ctx->ev_done[k][j]);
    }
    if (ctx->ev_ext) (void)hipEventDestroy(ctx->ev_ext);
    for (int k = 0; k < kSets; k++)
        if (ctx->ev_upd[k]) (void)hipEventDestroy(ctx->ev_upd[k]);
    if (ctx->ev_staged) (void)hipEventDestroy(ctx->ev_staged);
    if (ctx->ev_taken) (void)hipEventDestroy(ctx->ev_taken);
    if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
    (void)hipFree(ctx->d_upd);
    for (int k = 1; k < kSets; k++)
        if (ctx->fstream[k]) (void)hipStreamDestroy(ctx->fstream[k]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *ar_last_error(ar_ctx *ctx) { return ctx ? ctx->err : g_create_error; }

int ar_set_mesh(ar_ctx *ctx, const float *vbo9, uint32_t n_verts, const uint32_t *idx, uint32_t n_tris) {
    if (!ctx) return AR_E_ARG;
    if (n_tris > 0x7FFFFFFFu) return fail(ctx, AR_E_ARG, "ar_set_mesh: too many triangles (%u)", n_tris);
    if (n_tris && (!vbo9 || !idx || !n_verts)) return fail(ctx, AR_E_ARG, "ar_set_mesh: null buffer with %u triangles", n_tris);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = drop_mesh(ctx);
    if (rc || !n_tris) return rc;
    AR_HIP(ctx, hipMalloc(&ctx->d_vbo, size_t(n_verts) * 9 * sizeof(float)));
    AR_HIP(ctx, hipMalloc(&ctx->d_idx, size_t(n_tris) * 3 * sizeof(uint32_t)));
    AR_HIP(ctx, hipMemcpy(ctx->d_vbo, vbo9, size_t(n_verts) * 9 * sizeof(float), hipMemcpyHostToDevice));
    AR_HIP(ctx, hipMemcpy(ctx->d_idx, idx, size_t(n_tris) * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
    return finish_mesh(ctx, n_verts, n_tris);
}

int ar_set_mesh_indexed(ar_ctx *ctx, const float *positions, uint32_t n_verts, const uint32_t *indices, uint32_t n_tris,
                        const float *tri_colors) {
    if (!ctx) return AR_E_ARG;
    if (n_tris > 0x7FFFFFFFu / 3u) return fail(ctx, AR_E_ARG, "ar_set_mesh_indexed: too many triangles (%u)", n_tris);
    if (n_tris && (!positions || !indices || !n_verts))
        return fail(ctx, AR_E_ARG, "ar_set_mesh_indexed: null buffer with %u triangles", n_tris);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = drop_mesh(ctx);
    if (rc || !n_tris) return rc;
    float *d_pos = nullptr, *d_col = nullptr;
    uint32_t *d_ind = nullptr;
    auto staged = [&]() -> int {
        AR_HIP(ctx, hipMalloc(&d_pos, size_t(n_verts) * 3 * sizeof(float)));
        AR_HIP(ctx, hipMalloc(&d_ind, size_t(n_tris) * 3 * sizeof(uint32_t)));
        if (tri_colors) AR_HIP(ctx, hipMalloc(&d_col, size_t(n_tris) * 3 * sizeof(float)));
        AR_HIP(ctx, hipMalloc(&ctx->d_vbo, size_t(n_tris) * 27 * sizeof(float)));
        AR_HIP(ctx, hipMalloc(&ctx->d_idx, size_t(n_tris) * 3 * sizeof(uint32_t)));
        AR_HIP(ctx, hipMemcpy(d_pos, positions, size_t(n_verts) * 3 * sizeof(float), hipMemcpyHostToDevice));
        AR_HIP(ctx, hipMemcpy(d_ind, indices, size_t(n_tris) * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (tri_colors) AR_HIP(ctx, hipMemcpy(d_col, tri_colors, size_t(n_tris) * 3 * sizeof(float), hipMemcpyHostToDevice));
        k_build_vbo<<<(n_tris + 255) / 256, 256, 0, ctx->stream>>>(d_pos, n_verts, d_ind, n_tris, d_col, ctx->d_vbo, ctx->d_idx);
        AR_HIP(ctx, hipGetLastError());
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    rc = staged();
    (void)hipFree(d_pos); (void)hipFree(d_ind); (void)hipFree(d_col);
    if (rc) return rc;
    return finish_mesh(ctx, 3u * n_tris, n_tris);
}

int ar_update_positions(ar_ctx *ctx, const float *data, uint32_t n_verts, int floats_per_vertex, int flags) {
    if (!ctx) return AR_E_ARG;
    if (!data || (floats_per_vertex != 3 && floats_per_vertex != 9) || (flags & ~(AR_UPDATE_FLAT_NORMALS | AR_UPDATE_RESORT)))
        return fail(ctx, AR_E_ARG, "ar_update_positions: bad arguments (%d floats per vertex, flags %d)", floats_per_vertex, flags);
    if (!ctx->n_tris || !ctx->d_vbo) return fail(ctx, AR_E_STATE, "ar_update_positions: no mesh to update");
    if (n_verts != ctx->n_verts) return fail(ctx, AR_E_ARG, "ar_update_positions: %u vertices for a mesh of %u", n_verts, ctx->n_verts);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n_tris = ctx->n_tris;
    hipStream_t s = ctx->own_stream;  // == fstream[0]
    const size_t n_floats = size_t(n_verts) * size_t(floats_per_vertex);
    if (ctx->upd_floats < n_floats) {  // (first update of this size only)
        int rc = sync_streams(ctx);
        if (rc) return rc;
        (void)hipFree(ctx->d_upd);
        ctx->d_upd = nullptr; ctx->upd_floats = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_upd, n_floats * sizeof(float)));
        ctx->upd_floats = n_floats;
    }
    // The caller's data goes to a staging buffer on a stream of its own, and the host waits for that copy alone (so
    // `data` is free again on return) - behind the previous update's use of the buffer, not behind any frame.
    if (ctx->upd_pending) AR_HIP(ctx, hipStreamWaitEvent(ctx->up_stream, ctx->ev_taken, 0));
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_upd, data, n_floats * sizeof(float), hipMemcpyHostToDevice, ctx->up_stream));
    AR_HIP(ctx, hipEventRecord(ctx->ev_staged, ctx->up_stream));
    AR_HIP(ctx, hipStreamSynchronize(ctx->up_stream));
    // frames in flight read the streams this call rewrites: the update waits for them on the GPU, not on the host
    for (int k = 1; k < kSets; k++) {
        AR_HIP(ctx, hipEventRecord(ctx->ev_upd[k], ctx->fstream[k]));
        AR_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_upd[k], 0));
    }
    if (ctx->stream != ctx->own_stream) {
        AR_HIP(ctx, hipEventRecord(ctx->ev_upd[0], ctx->stream));
        AR_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_upd[0], 0));
    }
    AR_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_staged, 0));
    if (floats_per_vertex == 9) AR_HIP(ctx, hipMemcpyAsync(ctx->d_vbo, ctx->d_upd, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
    else k_put_positions<<<(n_verts + 255) / 256, 256, 0, s>>>(ctx->d_upd, n_verts, ctx->d_vbo);
    AR_HIP(ctx, hipEventRecord(ctx->ev_taken, s));
    ctx->upd_pending = true;
    if (flags & AR_UPDATE_FLAT_NORMALS) k_flat_normals<<<(n_tris + 255) / 256, 256, 0, s>>>(ctx->d_vbo, ctx->d_idx, n_verts, n_tris);
    AR_HIP(ctx, hipGetLastError());
    if (flags & AR_UPDATE_RESORT) {
        // a new spatial order: what ar_set_mesh does after its upload (allocates, and waits for the GPU)
        int rc = sync_streams(ctx);
        if (rc) return rc;
        const bool had_uv = ctx->d_triuv != nullptr;
        (void)hipFree(ctx->d_tripos); (void)hipFree(ctx->d_triattr); (void)hipFree(ctx->d_tri); (void)hipFree(ctx->d_tri_t);
        (void)hipFree(ctx->d_perm); (void)hipFree(ctx->d_box);
        ctx->d_tripos = nullptr; ctx->d_triattr = nullptr; ctx->d_tri = nullptr; ctx->d_tri_t = nullptr; ctx->d_perm = nullptr; ctx->d_box = nullptr;
        rc = finish_mesh(ctx, n_verts, n_tris);
        if (rc) return rc;
        if (had_uv) {
            k_uvflags<<<(n_tris + 255) / 256, 256, 0, s>>>(ctx->d_triuv, ctx->d_perm, n_tris, ctx->d_tri, ctx->d_tri_t);
            AR_HIP(ctx, hipGetLastError());
            AR_HIP(ctx, hipStreamSynchronize(s));
        }
        return 0;
    }
    // the derived streams, rewritten in the spatial order chosen at upload (slot -> triangle stays as it is)
    const unsigned blocks = unsigned((size_t(n_tris) + 255) / 256);
    const bool attr = floats_per_vertex == 9 || (flags & AR_UPDATE_FLAT_NORMALS);  // normals or colours may have changed too
    if (attr) k_tristream<true><<<blocks, 256, 0, s>>>(ctx->d_vbo, ctx->d_idx, ctx->d_perm, n_verts, n_tris, ctx->tripos_stride,
                                                      ctx->d_tripos, ctx->d_triattr, ctx->d_tri, ctx->d_tri_t, ctx->d_box);
    else k_tristream<false><<<blocks, 256, 0, s>>>(ctx->d_vbo, ctx->d_idx, ctx->d_perm, n_verts, n_tris, ctx->tripos_stride,
                                                  ctx->d_tripos, ctx->d_triattr, ctx->d_tri, ctx->d_tri_t, ctx->d_box);
    if (attr && ctx->d_triuv) k_uvflags<<<(n_tris + 255) / 256, 256, 0, s>>>(ctx->d_triuv, ctx->d_perm, n_tris, ctx->d_tri, ctx->d_tri_t);
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipEventRecord(ctx->ev_upd[0], s));
    for (int k = 1; k < kSets; k++) AR_HIP(ctx, hipStreamWaitEvent(ctx->fstream[k], ctx->ev_upd[0], 0));
    if (ctx->stream != ctx->own_stream) AR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upd[0], 0));
    return 0;
}

int ar_read_mesh(ar_ctx *ctx, float *vbo9, uint32_t *idx, uint32_t *n_verts, uint32_t *n_tris) {
    if (!ctx) return AR_E_ARG;
    if (n_verts) *n_verts = ctx->n_verts;
    if (n_tris) *n_tris = ctx->n_tris;
    if (!ctx->n_tris || (!vbo9 && !idx)) return 0;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    if (vbo9) AR_HIP(ctx, hipMemcpy(vbo9, ctx->d_vbo, size_t(ctx->n_verts) * 9 * sizeof(float), hipMemcpyDeviceToHost));
    if (idx) AR_HIP(ctx, hipMemcpy(idx, ctx->d_idx, size_t(ctx->n_tris) * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int ar_set_mesh_uv(ar_ctx *ctx, const uint16_t *uv8, uint32_t n_tris) {
    if (!ctx) return AR_E_ARG;
    if (uv8 && n_tris != ctx->n_tris)
        return fail(ctx, AR_E_ARG, "ar_set_mesh_uv: %u rows for a mesh of %u triangles", n_tris, ctx->n_tris);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    (void)hipFree(ctx->d_triuv);
    ctx->d_triuv = nullptr;
    if (!ctx->n_tris) return 0;
    if (uv8) {
        AR_HIP(ctx, hipMalloc(&ctx->d_triuv, size_t(n_tris) * sizeof(uint4)));
        AR_HIP(ctx, hipMemcpy(ctx->d_triuv, uv8, size_t(n_tris) * sizeof(uint4), hipMemcpyHostToDevice));
    }
    k_uvflags<<<(ctx->n_tris + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_triuv, ctx->d_perm, ctx->n_tris, ctx->d_tri, ctx->d_tri_t);
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int ar_set_texture(ar_ctx *ctx, const uint8_t *rgba, int width, int height) {
    if (!ctx) return AR_E_ARG;
    if (rgba && (width < 1 || height < 1 || width > 16384 || height > 16384))
        return fail(ctx, AR_E_ARG, "ar_set_texture: bad atlas size %d x %d", width, height);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    (void)hipFree(ctx->d_atlas);
    ctx->d_atlas = nullptr;
    ctx->atlas_w = ctx->atlas_h = 0;
    if (!rgba) return 0;
    AR_HIP(ctx, hipMalloc(&ctx->d_atlas, size_t(width) * height * 4));
    AR_HIP(ctx, hipMemcpy(ctx->d_atlas, rgba, size_t(width) * height * 4, hipMemcpyHostToDevice));
    ctx->atlas_w = width; ctx->atlas_h = height;
    return 0;
}

int ar_set_camera(ar_ctx *ctx, const float proj[16], const float view[16]) {
    if (!ctx) return AR_E_ARG;
    if (!proj || !view) return fail(ctx, AR_E_ARG, "ar_set_camera: null matrix");
    memcpy(ctx->proj, proj, 16 * sizeof(float));
    memcpy(ctx->view, view, 16 * sizeof(float));
    return 0;
}

int ar_set_model(ar_ctx *ctx, const float model[16]) {
    if (!ctx) return AR_E_ARG;
    ctx->has_model = model != nullptr;
    if (model) memcpy(ctx->model, model, 16 * sizeof(float));
    return 0;
}

int ar_set_lights(ar_ctx *ctx, const float amb[3], const float dir[3], const float dircol[3], int n_pl,
                  const float *plpos, const float *plcol) {
    if (!ctx) return AR_E_ARG;
    if (!amb || !dir || !dircol) return fail(ctx, AR_E_ARG, "ar_set_lights: null vector");
    if (n_pl < 0 || n_pl > AR_MAX_POINT_LIGHTS || (n_pl && (!plpos || !plcol)))
        return fail(ctx, AR_E_ARG, "ar_set_lights: bad point-light count %d", n_pl);
    FrameParams &fp = ctx->fp;
    for (int k = 0; k < 3; k++) { fp.amb[k] = amb[k]; fp.ldir[k] = dir[k]; fp.lcol[k] = dircol[k]; }
    memset(fp.plpos, 0, sizeof fp.plpos);
    memset(fp.plcol, 0, sizeof fp.plcol);
    for (int k = 0; k < 3 * n_pl; k++) { fp.plpos[k] = plpos[k]; fp.plcol[k] = plcol[k]; }
    fp.n_pl = n_pl;
    return 0;
}

int ar_set_ascii(ar_ctx *ctx, const uint8_t *ramp, int ramp_len, int mode_on, int mode_radius, int mode_thresh) {
    if (!ctx) return AR_E_ARG;
    if (!ramp || ramp_len < 1 || ramp_len > AR_MAX_RAMP) return fail(ctx, AR_E_ARG, "ar_set_ascii: ramp length %d not in 1..%d", ramp_len, AR_MAX_RAMP);
    FrameParams &fp = ctx->fp;
    memset(fp.ramp, 32, sizeof fp.ramp);
    memcpy(fp.ramp, ramp, size_t(ramp_len));
    fp.ramp_len = ramp_len;
    fp.mode_on = mode_on ? 1 : 0;
    fp.mode_radius = mode_radius;
    fp.mode_thresh = mode_thresh;
    return 0;
}

int ar_set_strip(ar_ctx *ctx, int row0, int row1) {
    if (!ctx) return AR_E_ARG;
    if (row0 < 0 || row1 < row0) return fail(ctx, AR_E_ARG, "ar_set_strip: bad rows [%d,%d)", row0, row1);
    ctx->strip0 = row0; ctx->strip1 = row1;
    return 0;
}

int ar_render_async(ar_ctx *ctx, int cols, int rows) {
    if (!ctx) return AR_E_ARG;
    return enqueue_frame(ctx, cols, rows, true);
}

int ar_render_batch_async(ar_ctx *ctx, int cols, int rows, int n_frames, const float *cameras, size_t ascii_stride) {
    if (!ctx) return AR_E_ARG;
    return enqueue_batch(ctx, cols, rows, n_frames, cameras, ascii_stride);
}

int ar_read_batch(ar_ctx *ctx, int frame, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids) {
    if (!ctx) return AR_E_ARG;
    if (ctx->last_batch < 0 || frame < 0 || frame >= ctx->last_batch_frames)
        return fail(ctx, AR_E_STATE, "ar_read_batch: frame %d of a last batch of %d", frame, ctx->last_batch < 0 ? 0 : ctx->last_batch_frames);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    const BatchSet &b = ctx->batch[ctx->last_batch];
    const size_t n = size_t(ctx->batch_cols) * ctx->batch_rows, f = size_t(frame);
    if (ids && !b.d_ids) return fail(ctx, AR_E_STATE, "ar_read_batch: ids wanted but ar_set_id_output was not on for the batch");
    if (rgba) AR_HIP(ctx, hipMemcpy(rgba, b.d_rgba + f * n * 4, n * 4, hipMemcpyDeviceToHost));
    if (depth) AR_HIP(ctx, hipMemcpy(depth, b.d_depth + f * n, n * sizeof(float), hipMemcpyDeviceToHost));
    if (ascii) AR_HIP(ctx, hipMemcpy(ascii, b.d_ascii + f * n, n, hipMemcpyDeviceToHost));
    if (ids) AR_HIP(ctx, hipMemcpy(ids, b.d_ids + f * n, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

void *ar_batch_device_buffer(ar_ctx *ctx, int which, int frame) {
    if (!ctx || ctx->last_batch < 0 || frame < 0 || frame >= ctx->last_batch_frames) return nullptr;
    const BatchSet &b = ctx->batch[ctx->last_batch];
    const size_t n = size_t(ctx->batch_cols) * ctx->batch_rows, f = size_t(frame);
    switch (which) {
        case 0: return b.d_rgba + f * n * 4;
        case 1: return b.d_depth + f * n;
        case 2: return b.d_ascii + f * n;
        default: return nullptr;
    }
}

int ar_sync(ar_ctx *ctx) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    return sync_streams(ctx);
}

namespace {

// Copies the planes of the most recent frame that are wanted to host memory, on the stream that frame ran on (so the
// copies queue behind its kernels without a wait of their own), then waits once.
int read_outputs(ar_ctx *ctx, hipStream_t s, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids) {
    const size_t n = size_t(ctx->cols) * ctx->rows;
    const int o = ctx->last_oset;
    if (ids && !ctx->d_ids[o]) return fail(ctx, AR_E_STATE, "ar_read: ids wanted but ar_set_id_output was not on for the last frame");
    hipError_t e = hipSuccess;
    if (rgba && e == hipSuccess) e = hipMemcpyAsync(rgba, ctx->d_rgba[o], n * 4, hipMemcpyDeviceToHost, s);
    if (depth && e == hipSuccess) e = hipMemcpyAsync(depth, ctx->d_depth[o], n * sizeof(float), hipMemcpyDeviceToHost, s);
    if (ascii && e == hipSuccess) e = hipMemcpyAsync(ascii, ctx->d_ascii[o], n, hipMemcpyDeviceToHost, s);
    if (ids && e == hipSuccess) e = hipMemcpyAsync(ids, ctx->d_ids[o], n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    const hipError_t done = hipStreamSynchronize(s);  // also after a failure: no copy into the caller's memory may outlive the call
    if (e != hipSuccess) return fail(ctx, AR_E_HIP, "device-to-host copy failed: %s", hipGetErrorString(e));
    if (done != hipSuccess) return fail(ctx, AR_E_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(done));
    return 0;
}

}  // namespace

int ar_read(ar_ctx *ctx, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids) {
    if (!ctx) return AR_E_ARG;
    if (!ctx->rendered) return fail(ctx, AR_E_STATE, "ar_read: no frame has been rendered at this grid size");
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    return read_outputs(ctx, ctx->stream, rgba, depth, ascii, ids);
}

int ar_render(ar_ctx *ctx, int cols, int rows, uint8_t *rgba, float *depth, uint8_t *ascii) {
    if (!ctx) return AR_E_ARG;
    int rc = enqueue_frame(ctx, cols, rows, false);
    if (rc) return rc;
    if (ctx->n_tris == 0 || !ctx->rendered) return ar_sync(ctx);  // no geometry: outputs stay untouched (raster.js:274)
    // a blocking frame runs on ctx->stream alone (enqueue_frame has waited for any overlapped frames before it)
    return read_outputs(ctx, ctx->stream, rgba, depth, ascii, nullptr);
}

int ar_read_async(ar_ctx *ctx, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids, uint64_t *ticket) {
    if (!ctx) return AR_E_ARG;
    if (!ticket) return fail(ctx, AR_E_ARG, "ar_read_async: null ticket");
    if (!ctx->rendered || !ctx->last_stream) return fail(ctx, AR_E_STATE, "ar_read_async: no frame has been rendered at this grid size");
    AR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = size_t(ctx->cols) * ctx->rows;
    const int o = ctx->last_oset, slot = ctx->last_slot;
    hipStream_t s = ctx->last_stream;
    if (ids && !ctx->d_ids[o]) return fail(ctx, AR_E_STATE, "ar_read_async: ids wanted but ar_set_id_output was not on for the last frame");
    hipError_t e = hipSuccess;
    if (rgba && e == hipSuccess) e = hipMemcpyAsync(rgba, ctx->d_rgba[o], n * 4, hipMemcpyDeviceToHost, s);
    if (depth && e == hipSuccess) e = hipMemcpyAsync(depth, ctx->d_depth[o], n * sizeof(float), hipMemcpyDeviceToHost, s);
    if (ascii && e == hipSuccess) e = hipMemcpyAsync(ascii, ctx->d_ascii[o], n, hipMemcpyDeviceToHost, s);
    if (ids && e == hipSuccess) e = hipMemcpyAsync(ids, ctx->d_ids[o], n * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    const uint64_t serial = ctx->slot_serial[slot] + 1;
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_done[slot][serial % kTicketRing], s);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(s);  // no copy into the caller's memory may outlive a failed call
        return fail(ctx, AR_E_HIP, "ar_read_async: %s", hipGetErrorString(e));
    }
    ctx->slot_serial[slot] = serial;
    *ticket = (serial << 2) | uint64_t(slot);
    return 0;
}

namespace {

// The event behind a ticket, or null (message set) when the ticket was never issued by this context.  A slot keeps the
// events of its last kTicketRing tickets apart; a ticket older than that waits for the slot's newest event, which was
// recorded later on the same stream and therefore covers it.
hipEvent_t ticket_event(ar_ctx *ctx, uint64_t ticket, const char *who) {
    const uint64_t serial = ticket >> 2;
    const int slot = int(ticket & 3u);
    if (slot > kSets || serial == 0 || ctx->slot_serial[slot] < serial) {
        fail(ctx, AR_E_ARG, "%s: ticket %llu was not issued by this context", who, static_cast<unsigned long long>(ticket));
        return nullptr;
    }
    const uint64_t newest = ctx->slot_serial[slot];
    return ctx->ev_done[slot][(newest - serial < uint64_t(kTicketRing) ? serial : newest) % kTicketRing];
}

}  // namespace

int ar_wait(ar_ctx *ctx, uint64_t ticket) {
    if (!ctx) return AR_E_ARG;
    hipEvent_t ev = ticket_event(ctx, ticket, "ar_wait");
    if (!ev) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    AR_HIP(ctx, hipEventSynchronize(ev));
    return 0;
}

int ar_poll(ar_ctx *ctx, uint64_t ticket) {
    if (!ctx) return AR_E_ARG;
    hipEvent_t ev = ticket_event(ctx, ticket, "ar_poll");
    if (!ev) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(ctx, AR_E_HIP, "ar_poll: %s", hipGetErrorString(e));
}

void *ar_device_buffer(ar_ctx *ctx, int which) {
    if (!ctx) return nullptr;
    switch (which) {
        case 0: return ctx->d_rgba[ctx->last_oset];
        case 1: return ctx->d_depth[ctx->last_oset];
        case 2: return ctx->d_ascii[ctx->last_oset];
        default: return nullptr;
    }
}

void *ar_stream(ar_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

void *ar_host_alloc(ar_ctx *ctx, size_t bytes) {
    if (!ctx) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(ctx, AR_E_NOMEM, "ar_host_alloc: cannot allocate %zu bytes of page-locked host memory", bytes);
        return nullptr;
    }
    return p;
}

void ar_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int ar_set_stream(ar_ctx *ctx, void *hip_stream) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    ctx->ext_stream = hip_stream != nullptr;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return 0;
}

int ar_set_ascii_target(ar_ctx *ctx, void *d_ascii, int first_row) {
    if (!ctx) return AR_E_ARG;
    if (first_row < 0) return fail(ctx, AR_E_ARG, "ar_set_ascii_target: bad first row %d", first_row);
    // (alignment is checked per frame, where the row length is known: see ascii_target_misaligned)
    ctx->ascii_target = static_cast<uint8_t *>(d_ascii);
    ctx->target_row = d_ascii ? first_row : 0;
    return 0;
}

int ar_stream_wait_frame(ar_ctx *ctx, void *hip_stream) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    for (int k = 0; k < kSets; k++) {
        AR_HIP(ctx, hipEventRecord(ctx->ev_out[k], ctx->fstream[k]));
        AR_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_out[k], 0));
    }
    if (ctx->stream != ctx->own_stream && ctx->stream != s) {
        AR_HIP(ctx, hipEventRecord(ctx->ev_out[kSets], ctx->stream));
        AR_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_out[kSets], 0));
    }
    return 0;
}

int ar_frame_wait_stream(ar_ctx *ctx, void *hip_stream) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    AR_HIP(ctx, hipEventRecord(ctx->ev_ext, static_cast<hipStream_t>(hip_stream)));
    for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipStreamWaitEvent(ctx->fstream[k], ctx->ev_ext, 0));
    if (ctx->stream != ctx->own_stream) AR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_ext, 0));
    return 0;
}

namespace {

// Host RGBA grid -> d_q_rgba and its ASCII codes -> d_q_ascii, enqueued on the context's stream.
int quantize_to_device(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows) {
    AR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = size_t(cols) * rows;
    if (n > ctx->q_cells) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii); (void)hipFree(ctx->d_q_plane);
        ctx->d_q_rgba = nullptr; ctx->d_q_ascii = nullptr; ctx->d_q_plane = nullptr; ctx->q_cells = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_q_rgba, n * 4));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_ascii, n));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_plane, n));
        ctx->q_cells = n;
    }
    FrameParams fp = ctx->fp;
    fp.cols = cols; fp.rows = rows; fp.row0 = 0; fp.row1 = rows; fp.ya = 0; fp.yb = rows;
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_q_rgba, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        Stamper s(ctx, K_ASCII);
        if (fp.mode_on == 1) launch_ascii_rgba(fp, reinterpret_cast<const uint32_t *>(ctx->d_q_rgba), ctx->d_q_plane, false, ctx->d_q_ascii, ctx->stream);
        else k_quantize_plain<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(fp, reinterpret_cast<const uint32_t *>(ctx->d_q_rgba), ctx->d_q_ascii, n);
    }
    AR_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace

int ar_quantize(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, uint8_t *ascii) {
    if (!ctx) return AR_E_ARG;
    if (!rgba || !ascii || cols <= 0 || rows <= 0 || size_t(cols) * rows > (size_t(1) << 31))
        return fail(ctx, AR_E_ARG, "ar_quantize: bad arguments (%d x %d)", cols, rows);
    int rc = quantize_to_device(ctx, rgba, cols, rows);
    if (rc) return rc;
    AR_HIP(ctx, hipMemcpyAsync(ascii, ctx->d_q_ascii, size_t(cols) * rows, hipMemcpyDeviceToHost, ctx->stream));
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ar_quantize through the static staging array (defined above ar_create): no hipMalloc / hipFree on this path.
int ar_quantize_static(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, uint8_t *ascii) {
    if (!ctx) return AR_E_ARG;
    if (!rgba || !ascii || cols <= 0 || rows <= 0) return fail(ctx, AR_E_ARG, "ar_quantize_static: bad arguments (%d x %d)", cols, rows);
    const size_t n = size_t(cols) * size_t(rows);
    if (n > size_t(AR_STATIC_MAX_CELLS))
        return fail(ctx, AR_E_ARG, "ar_quantize_static: %d x %d = %zu cells exceed the static staging array's %d cells (use ar_quantize)",
                    cols, rows, n, AR_STATIC_MAX_CELLS);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    unsigned char *base = nullptr;
    int rc = static_scratch(ctx, &base);  // (normally found at ar_create; looked up again here if that failed, and reported here)
    if (rc) return rc;
    uint8_t *d_rgba = base, *d_ascii = base + size_t(AR_STATIC_MAX_CELLS) * 4, *d_plane = base + size_t(AR_STATIC_MAX_CELLS) * 5;
    FrameParams fp = ctx->fp;
    fp.cols = cols; fp.rows = rows; fp.row0 = 0; fp.row1 = rows; fp.ya = 0; fp.yb = rows;
    AR_HIP(ctx, hipMemcpyAsync(d_rgba, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (fp.mode_on == 1) launch_ascii_rgba(fp, reinterpret_cast<const uint32_t *>(d_rgba), d_plane, false, d_ascii, ctx->stream);
    else k_quantize_plain<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(fp, reinterpret_cast<const uint32_t *>(d_rgba), d_ascii, n);
    hipError_t launched = hipGetLastError();
    if (launched == hipSuccess) launched = hipMemcpyAsync(ascii, d_ascii, n, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t done = hipStreamSynchronize(ctx->stream);  // also on failure: the copy of the caller's buffer must not outlive the call
    if (launched != hipSuccess) return fail(ctx, AR_E_HIP, "ar_quantize_static: %s", hipGetErrorString(launched));
    if (done != hipSuccess) return fail(ctx, AR_E_HIP, "ar_quantize_static: %s", hipGetErrorString(done));
    return 0;
}

int ar_rt_set_scene(ar_ctx *ctx, const ar_rt_scene *sc) {
    if (!ctx) return AR_E_ARG;
    if (!sc) return fail(ctx, AR_E_ARG, "ar_rt_set_scene: null scene");
    const int counts[6] = {sc->n_spheres, sc->n_planes, sc->n_tris, sc->n_mats, sc->n_pl, sc->n_dl};
    for (int c : counts)
        if (c < 0 || c > 65536) return fail(ctx, AR_E_ARG, "ar_rt_set_scene: count %d out of range", c);
    if ((sc->n_spheres && (!sc->spheres4 || !sc->sphere_mat)) || (sc->n_planes && (!sc->planes4 || !sc->plane_mat)) ||
        (sc->n_tris && (!sc->tri_a || !sc->tri_b || !sc->tri_c || !sc->tri_mat)) || (sc->n_mats && (!sc->mat_albedo || !sc->mat_reflective)) ||
        (sc->n_pl && (!sc->pl_pos || !sc->pl_col)) || (sc->n_dl && (!sc->dl_dir || !sc->dl_col)))
        return fail(ctx, AR_E_ARG, "ar_rt_set_scene: null array with a non-zero count");
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    // one host image of all arrays (4-byte elements), one allocation, one copy
    std::vector<uint32_t> blob;
    auto put = [&](const void *src, size_t words) {
        const size_t at = blob.size();
        blob.resize(at + (words ? words : 1));
        if (words) memcpy(blob.data() + at, src, words * 4);
        return at;
    };
    const size_t o_s4 = put(sc->spheres4, size_t(sc->n_spheres) * 4), o_sm = put(sc->sphere_mat, size_t(sc->n_spheres));
    const size_t o_p4 = put(sc->planes4, size_t(sc->n_planes) * 4), o_pm = put(sc->plane_mat, size_t(sc->n_planes));
    const size_t o_ta = put(sc->tri_a, size_t(sc->n_tris) * 3), o_tb = put(sc->tri_b, size_t(sc->n_tris) * 3);
    const size_t o_tc = put(sc->tri_c, size_t(sc->n_tris) * 3), o_tm = put(sc->tri_mat, size_t(sc->n_tris));
    const size_t o_ma = put(sc->mat_albedo, size_t(sc->n_mats) * 3), o_mr = put(sc->mat_reflective, size_t(sc->n_mats));
    const size_t o_pp = put(sc->pl_pos, size_t(sc->n_pl) * 3), o_pc = put(sc->pl_col, size_t(sc->n_pl) * 3);
    const size_t o_dd = put(sc->dl_dir, size_t(sc->n_dl) * 3), o_dc = put(sc->dl_col, size_t(sc->n_dl) * 3);
    (void)hipFree(ctx->d_rt);
    ctx->d_rt = nullptr;
    ctx->rt_has_scene = false;
    AR_HIP(ctx, hipMalloc(&ctx->d_rt, blob.size() * 4));
    AR_HIP(ctx, hipMemcpy(ctx->d_rt, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    const float *f = static_cast<const float *>(ctx->d_rt);
    const int32_t *w = static_cast<const int32_t *>(ctx->d_rt);
    RtScene &S = ctx->rt_scene;
    S.n_spheres = sc->n_spheres; S.n_planes = sc->n_planes; S.n_tris = sc->n_tris; S.n_mats = sc->n_mats; S.n_pl = sc->n_pl; S.n_dl = sc->n_dl;
    S.spheres4 = f + o_s4; S.sphere_mat = w + o_sm; S.planes4 = f + o_p4; S.plane_mat = w + o_pm;
    S.tri_a = f + o_ta; S.tri_b = f + o_tb; S.tri_c = f + o_tc; S.tri_mat = w + o_tm;
    S.mat_albedo = f + o_ma; S.mat_reflective = w + o_mr;
    S.pl_pos = f + o_pp; S.pl_col = f + o_pc; S.dl_dir = f + o_dd; S.dl_col = f + o_dc;
    for (int k = 0; k < 3; k++) S.env[k] = sc->env[k];
    ctx->rt_has_scene = true;
    return 0;
}

int ar_rt_render_async(ar_ctx *ctx, int cols, int rows, const float cam_pos[3], float yaw, float pitch, float fovy, float pix_aspect) {
    if (!ctx) return AR_E_ARG;
    if (!cam_pos || cols <= 0 || rows <= 0 || size_t(cols) * rows > (size_t(1) << 31))
        return fail(ctx, AR_E_ARG, "ar_rt_render: bad arguments (%d x %d)", cols, rows);
    if (ascii_target_misaligned(ctx, cols))
        return fail(ctx, AR_E_ARG, "ar_rt_render: the ASCII target must be 4-byte aligned when the row length (%d) is a multiple of four", cols);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_targets(ctx, cols, rows);
    if (rc) return rc;
    if (!ctx->rt_has_scene) return 0;  // raytrace.js:321: no scene, nothing drawn
    if (ctx->overlapped_in_flight) {   // raster frames may still be writing output set 0
        rc = sync_streams(ctx);
        if (rc) return rc;
    }
    FrameParams fp = ctx->fp;
    fp.cols = cols; fp.rows = rows;
    int row0 = 0, row1 = rows;
    if (ctx->strip1 > ctx->strip0) {
        if (ctx->strip1 > rows) return fail(ctx, AR_E_ARG, "strip [%d,%d) exceeds %d rows", ctx->strip0, ctx->strip1, rows);
        row0 = ctx->strip0; row1 = ctx->strip1;
    }
    const int halo = fp.mode_on == 1 ? 3 : 0;
    fp.row0 = row0; fp.row1 = row1;
    fp.ya = row0 - halo < 0 ? 0 : row0 - halo;
    fp.yb = row1 + halo > rows ? rows : row1 + halo;
    // camera terms of raytrace_shader.js:66-70,203-207: binary64 libm on the binary32 uniforms, rounded once
    RtCamera cam;
    const double cp = std::cos(double(pitch)), sp = std::sin(double(pitch)), cy = std::cos(double(yaw)), sy = std::sin(double(yaw));
    auto dot3 = [](const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
    auto norm3 = [&](float *v) {
        const float d = dot3(v, v);
        const float k = d > 0.0f ? 1.0f / std::sqrt(d) : 0.0f;
        if (d > 0.0f) { v[0] = v[0] * k; v[1] = v[1] * k; v[2] = v[2] * k; } else { v[0] = v[1] = v[2] = 0.0f; }
    };
    auto cross3 = [](const float *a, const float *b, float *o) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    cam.ww[0] = float(cp * cy); cam.ww[1] = float(sp); cam.ww[2] = float(cp * sy);
    norm3(cam.ww);
    const float up[3] = {0.0f, 1.0f, 0.0f};
    cross3(cam.ww, up, cam.uu);
    norm3(cam.uu);
    if (std::sqrt(dot3(cam.uu, cam.uu)) < 1e-3f) { cam.uu[0] = 1.0f; cam.uu[1] = 0.0f; cam.uu[2] = 0.0f; }
    cross3(cam.uu, cam.ww, cam.vv);
    norm3(cam.vv);
    cam.focal = 1.0f / std::fmax(1e-6f, float(std::tan(0.5 * double(fovy))));
    cam.aspect = (float(cols) / float(rows)) * std::fmax(pix_aspect, 1e-6f);
    for (int k = 0; k < 3; k++) cam.ro[k] = cam_pos[k];
    const size_t strip_cells = size_t(fp.yb - fp.ya) * cols;
    uint8_t *ascii_out = ctx->ascii_target ? ctx->ascii_target - ptrdiff_t(ctx->target_row) * cols : ctx->d_ascii[0];
    {
        Stamper s(ctx, K_RAYTRACE);
        k_raytrace<<<unsigned((strip_cells + 255) / 256), 256, 0, ctx->stream>>>(ctx->rt_scene, cam, cols, rows, fp.ya, fp.yb,
                                                                               reinterpret_cast<uint32_t *>(ctx->d_rgba[0]));
    }
    {
        Stamper s(ctx, K_ASCII);
        if (fp.mode_on == 1) {
            launch_ascii_rgba(fp, reinterpret_cast<const uint32_t *>(ctx->d_rgba[0]), ctx->d_ridx[0], false, ascii_out, ctx->stream);
        } else {
            const size_t first = size_t(row0) * cols, n = size_t(row1 - row0) * cols;
            k_quantize_plain<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(
                fp, reinterpret_cast<const uint32_t *>(ctx->d_rgba[0]) + first, ascii_out + first, n);
        }
    }
    AR_HIP(ctx, hipGetLastError());
    ctx->rendered = true;
    ctx->last_oset = 0;
    ctx->last_stream = ctx->stream;
    ctx->last_slot = kSets;
    return 0;
}

int ar_rt_render(ar_ctx *ctx, int cols, int rows, const float cam_pos[3], float yaw, float pitch, float fovy, float pix_aspect,
                 uint8_t *rgba, uint8_t *ascii) {
    int rc = ar_rt_render_async(ctx, cols, rows, cam_pos, yaw, pitch, fovy, pix_aspect);
    if (rc) return rc;
    if (!ctx->rt_has_scene) return 0;
    return ar_read(ctx, rgba, nullptr, ascii, nullptr);
}

int ar_present(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, const uint8_t *atlas_alpha, int cell_w, int cell_h,
               double alpha_gamma, int grayscale, int transparent_bg, uint8_t *image) {
    if (!ctx) return AR_E_ARG;
    if (!rgba || !atlas_alpha || !image || cols <= 0 || rows <= 0 || cell_w <= 0 || cell_h <= 0 || cell_w > 1024 || cell_h > 1024 ||
        size_t(cols) * rows > (size_t(1) << 31) || size_t(cols) * cell_w > (size_t(1) << 20) || size_t(rows) * cell_h > 65535u ||
        !(alpha_gamma > 0.0))
        return fail(ctx, AR_E_ARG, "ar_present: bad arguments (%d x %d cells of %d x %d pixels, gamma %g)", cols, rows, cell_w, cell_h, alpha_gamma);
    int rc = quantize_to_device(ctx, rgba, cols, rows);
    if (rc) return rc;
    PresentParams pp;
    pp.cols = cols; pp.rows = rows; pp.cell_w = cell_w; pp.cell_h = cell_h;
    pp.W = cols * cell_w; pp.H = rows * cell_h;
    pp.tile_w = cell_w + 4; pp.tile_h = cell_h + 4; pp.atlas_w = 16 * pp.tile_w;
    pp.grayscale = grayscale ? 1 : 0; pp.transparent_bg = transparent_bg ? 1 : 0; pp.vec_store = (pp.W % 4) == 0 ? 1 : 0;
    pp.inv_cell_w = 1.0f / float(cell_w);
    const size_t atlas_bytes = size_t(pp.atlas_w) * 16 * pp.tile_h, pixels = size_t(pp.W) * pp.H;
    if (atlas_bytes > ctx->p_atlas_bytes || !ctx->d_p_cov) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_p_atlas); ctx->d_p_atlas = nullptr; ctx->p_atlas_bytes = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_p_atlas, atlas_bytes));
        ctx->p_atlas_bytes = atlas_bytes;
        if (!ctx->d_p_cov) AR_HIP(ctx, hipMalloc(&ctx->d_p_cov, 256 * sizeof(float)));
    }
    if (pixels > ctx->p_pixels) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_p_image); ctx->d_p_image = nullptr; ctx->p_pixels = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_p_image, pixels * sizeof(uint32_t)));
        ctx->p_pixels = pixels;
    }
    float cov_of[256];
    for (int a = 0; a < 256; a++) cov_of[a] = alpha_gamma == 1.0 ? float(a) / 255.0f : float(std::pow(double(a) / 255.0, alpha_gamma));
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_p_atlas, atlas_alpha, atlas_bytes, hipMemcpyHostToDevice, ctx->stream));
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_p_cov, cov_of, sizeof cov_of, hipMemcpyHostToDevice, ctx->stream));
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // cov_of lives on this stack frame
    {
        Stamper s(ctx, K_PRESENT);
        k_present<<<dim3(unsigned((pp.W + 1023) / 1024), unsigned(pp.H)), 256, 0, ctx->stream>>>(
            pp, reinterpret_cast<const uint32_t *>(ctx->d_q_rgba), ctx->d_q_ascii, ctx->d_p_atlas, ctx->d_p_cov, ctx->d_p_image);
    }
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipMemcpyAsync(image, ctx->d_p_image, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

namespace {

// Math.round of a non-negative double (halves up), as JavaScript rounds in qChan (js/renderer.js:30).
double js_round(double x) {
    double r = std::floor(x);
    if (x - r >= 0.5) r += 1.0;
    return r;
}

// steps <= 0: the reference's max(3, max(2, ramp length)) (js/renderer.js:13-19); else checked against 2..256.
int resolve_color_steps(ar_ctx *ctx, int &steps) {
    if (steps <= 0) {
        const int rl = ctx->fp.ramp_len < 2 ? 2 : ctx->fp.ramp_len;
        steps = rl < 3 ? 3 : rl;
    }
    if (steps < 2 || steps > 256) return fail(ctx, AR_E_ARG, "colour steps %d not in 2..256", steps);
    return 0;
}

// d_rgba (device, n cells) -> colors (host).
int colors_from_device(ar_ctx *ctx, const uint8_t *d_rgba, size_t n, int steps, int grayscale, uint32_t *colors) {
    int rs = resolve_color_steps(ctx, steps);
    if (rs) return rs;
    if (grayscale) {  // black text everywhere (renderer.js:76-77)
        memset(colors, 0, n * sizeof(uint32_t));
        return 0;
    }
    ColorLut lut;
    const double step_val = 255.0 / double(steps - 1);
    for (int u = 0; u < 256; u++) lut.q[u] = uint8_t(js_round(js_round((double(u) / 255.0) * double(steps - 1)) * step_val));
    if (n > ctx->qc_cells) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_q_colors);
        ctx->d_q_colors = nullptr; ctx->qc_cells = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_q_colors, n * sizeof(uint32_t)));
        ctx->qc_cells = n;
    }
    k_colors<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(lut, reinterpret_cast<const uint32_t *>(d_rgba), ctx->d_q_colors, n);
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipMemcpyAsync(colors, ctx->d_q_colors, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace

int ar_quantize_colors(ar_ctx *ctx, const uint8_t *rgba, int cols, int rows, int steps, int grayscale, uint32_t *colors) {
    if (!ctx) return AR_E_ARG;
    if (!rgba || !colors || cols <= 0 || rows <= 0 || size_t(cols) * rows > (size_t(1) << 31))
        return fail(ctx, AR_E_ARG, "ar_quantize_colors: bad arguments (%d x %d)", cols, rows);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = size_t(cols) * rows;
    // the cases that need no device work return before anything of the caller's is handed to the copy engine
    int rc = resolve_color_steps(ctx, steps);
    if (rc) return rc;
    if (grayscale) return colors_from_device(ctx, nullptr, n, steps, grayscale, colors);
    if (n > ctx->q_cells) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii); (void)hipFree(ctx->d_q_plane);
        ctx->d_q_rgba = nullptr; ctx->d_q_ascii = nullptr; ctx->d_q_plane = nullptr; ctx->q_cells = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_q_rgba, n * 4));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_ascii, n));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_plane, n));
        ctx->q_cells = n;
    }
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_q_rgba, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = colors_from_device(ctx, ctx->d_q_rgba, n, steps, grayscale, colors);
    if (rc) (void)hipStreamSynchronize(ctx->stream);  // the copy of the caller's buffer must not outlive the call
    return rc;
}

int ar_read_colors(ar_ctx *ctx, int steps, int grayscale, uint32_t *colors) {
    if (!ctx) return AR_E_ARG;
    if (!colors) return fail(ctx, AR_E_ARG, "ar_read_colors: null output");
    if (!ctx->rendered) return fail(ctx, AR_E_STATE, "ar_read_colors: no frame has been rendered at this grid size");
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    return colors_from_device(ctx, ctx->d_rgba[ctx->last_oset], size_t(ctx->cols) * ctx->rows, steps, grayscale, colors);
}

int ar_set_id_output(ar_ctx *ctx, int on) {
    if (!ctx) return AR_E_ARG;
    ctx->want_ids = on != 0;
    return 0;
}

int ar_set_profiling(ar_ctx *ctx, int on) {
    if (!ctx) return AR_E_ARG;
    ctx->profiling = on != 0;
    return 0;
}

int ar_kernel_stats(ar_ctx *ctx, const char *names[AR_N_KERNELS], uint32_t launches[AR_N_KERNELS], float total_ms[AR_N_KERNELS]) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    for (int k = 0; k < AR_N_KERNELS; k++) {
        if (names) names[k] = kKernelNames[k];
        if (launches) launches[k] = 0;
        if (total_ms) total_ms[k] = 0.0f;
    }
    for (auto &s : ctx->stamps) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            if (launches) launches[s.slot] += 1;
            if (total_ms) total_ms[s.slot] += ms;
        }
        ctx->free_events.push_back(s.a);
        ctx->free_events.push_back(s.b);
    }
    ctx->stamps.clear();
    return 0;
}

int ar_frame_counters(ar_ctx *ctx, uint32_t out[8]) {
    if (!ctx || !out) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    uint32_t c[4] = {0, 0, 0, 0}, last[2] = {0, 0};  // last: {overflow, n_entries} of the last tiled frame
    AR_HIP(ctx, hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    AR_HIP(ctx, hipMemcpy(last, &ctx->d_state[(ctx->frame + unsigned(kSets) - 1u) % unsigned(kSets)].overflow, sizeof last, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = ctx->n_tris;
    out[1] = ctx->last_tiled ? 0 : c[2];
    out[2] = ctx->last_tiled ? 0 : c[3];
    out[3] = ctx->last_tiled ? last[1] : 0;
    out[4] = ctx->last_tiled ? last[0] : 0;
    out[5] = ctx->last_tiled ? 1u : 0u;
    return 0;
}

int ar_set_option(ar_ctx *ctx, int option, long long value) {
    if (!ctx) return AR_E_ARG;
    switch (option) {
        case AR_OPT_PATH:
            if (value != 0 && value != 1) return fail(ctx, AR_E_ARG, "AR_OPT_PATH takes 0 (tiled) or 1 (direct)");
            ctx->path = int(value);
            return 0;
        case AR_OPT_OVERLAP: {
            if (value != 0 && value != 1) return fail(ctx, AR_E_ARG, "AR_OPT_OVERLAP takes 0 or 1");
            int rc = sync_streams(ctx);
            if (rc) return rc;
            ctx->overlap = int(value);
            return 0;
        }
        case AR_OPT_POOL_RECORDS:
            if (value < 0 || value > 0x7FFFFFFFll) return fail(ctx, AR_E_ARG, "AR_OPT_POOL_RECORDS out of range");
            ctx->pool_cap_option = uint32_t(value);
            return 0;
        case AR_OPT_TILE_ROWS:
            if (value != 0 && value != 8 && value != 16 && value != 32) return fail(ctx, AR_E_ARG, "AR_OPT_TILE_ROWS takes 0, 8, 16 or 32");
            ctx->tile_shift_option = value == 8 ? 3 : value == 16 ? 4 : value == 32 ? 5 : 0;
            return 0;
        case AR_OPT_FOLD_SCAN:
            if (value != 0 && value != 1) return fail(ctx, AR_E_ARG, "AR_OPT_FOLD_SCAN takes 0 or 1");
            ctx->no_fold = value == 0;
            return 0;
        case AR_OPT_WORK_CLASSES:
            if (value != 0 && value != 1) return fail(ctx, AR_E_ARG, "AR_OPT_WORK_CLASSES takes 0 or 1");
            ctx->no_classes = value == 0;
            return 0;
        case AR_OPT_SMOOTHING_KERNEL:
            if (value < 0 || value > 3) return fail(ctx, AR_E_ARG, "AR_OPT_SMOOTHING_KERNEL takes 0 .. 3");
            ctx->ascii_path = int(value);
            ctx->fp.ascii_path = int(value);
            return 0;
        default:
            return fail(ctx, AR_E_ARG, "unknown option %d", option);
    }
}

}  // extern "C"

#include "group.h"
