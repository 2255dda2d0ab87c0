// context.h - the renderer context (ar_ctx), its buffers and streams, and the per-frame enqueue
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

thread_local char g_create_error[256] = "";

}  // namespace

constexpr int kTicketRing = 8;  // tickets per stream whose events are kept apart (older ones wait for a newer event of the same stream)

constexpr int kMaxBatchFrames = 256;  // frames one ar_render_batch_async call takes

// Buffers of a batch of frames rendered by one launch chain (ar_render_batch_async): everything a frame writes, n times
// over, frame after frame.  There are kSets of these, used by consecutive batches in rotation like the buffer sets of
// single frames, so that batches overlap on the frame streams the same way.
struct BatchSet {
    int frames = 0;                // frames the buffers hold
    size_t cells = 0;              // cells per frame
    size_t surv = 0, dircount = 0, dir_table = 0, entries = 0;  // per-frame sizes: survivor places, count-table words (+ totals), of which table, bin entries
    bool ids = false;
    uint2 *d_surv = nullptr; uint32_t *d_sid = nullptr, *d_surv_count = nullptr, *d_dircount = nullptr, *d_entries = nullptr;
    BinState *d_state = nullptr;
    uint64_t *d_vis = nullptr;
    uint8_t *d_ridx = nullptr, *d_rgba = nullptr, *d_ascii = nullptr;
    float *d_depth = nullptr;
    uint32_t *d_ids = nullptr;
    FrameParams *d_fps = nullptr, *h_fps = nullptr;  // the frames' parameters: device array, page-locked staging
    hipEvent_t ev_params = nullptr;                  // the staging array has been copied (it may be rewritten)
    bool params_pending = false;
    bool blocksum_zero = false;    // the totals behind every frame's count table are zero
};

struct ar_ctx {
    int device = 0;
    hipStream_t stream = nullptr;   // stream of frames that do not overlap, of quantize / present and of copies: own_stream
                                    // or the caller's (ar_set_stream)
    hipStream_t own_stream = nullptr;
    hipStream_t fstream[kSets] = {};  // overlapped frames: frame f runs entirely on fstream[f % kSets]; [0] is own_stream
    bool overlapped_in_flight = false;  // an overlapped frame was enqueued since the last full synchronisation
    bool ext_stream = false;        // stream is the caller's: every kernel goes there, frames do not overlap
    uint8_t *ascii_target = nullptr;  // caller's device buffer for the ASCII plane (ar_set_ascii_target), else d_ascii
    int target_row = 0;               // frame row stored at ascii_target
    int overlap = 1;                // 0: frames one after the other on `stream`; 1: consecutive frames rotate over the frame streams
    hipEvent_t ev_out[kSets + 1] = {};  // outputs complete on each frame stream / on `stream` (ar_stream_wait_frame)
    hipEvent_t ev_ext = nullptr;    // a caller's stream has consumed the outputs (ar_frame_wait_stream)
    // pipelined read-back (ar_read_async / ar_wait): a ring of events per frame stream and one for `stream`, each recorded
    // behind the copies of one frame; a ticket is (serial on its slot << 2 | slot), slot_serial[] the newest serial of a slot
    hipEvent_t ev_done[kSets + 1][kTicketRing] = {};
    uint64_t slot_serial[kSets + 1] = {};
    hipStream_t last_stream = nullptr;  // stream the most recent frame was enqueued on
    int last_slot = kSets;              // its event slot: the buffer set of an overlapped frame, kSets for `stream`
    float *d_vbo = nullptr;
    uint32_t *d_idx = nullptr;
    uint32_t n_verts = 0, n_tris = 0;
    float proj[16], view[16];
    float model[16];           // column-major model matrix (ar_set_model)
    bool has_model = false;
    FrameParams fp;
    int strip0 = 0, strip1 = 0;
    int cols = 0, rows = 0;  // size of the current targets
    uint64_t *d_vis[kSets] = {};  // global key buffers ([0] direct path; tiled path: overflow keys of the frame's set)
    // outputs, per buffer set while frames overlap (frames that do not overlap always use set 0)
    uint8_t *d_rgba[kSets] = {};
    float *d_depth[kSets] = {};
    uint8_t *d_ascii[kSets] = {};
    int last_oset = 0;           // output set of the most recent frame
    uint8_t *d_ridx[kSets] = {};  // ramp index per cell (input of the smoothing pass)
    uint32_t *d_ids[kSets] = {};  // winner per cell, allocated when id output is switched on
    bool want_ids = false;
    bool vis_clean = false;      // every key of d_vis[*] is "background" (the consuming kernels restore this each frame)
    TriRec *d_medium = nullptr, *d_large = nullptr;
    uint32_t *d_counters = nullptr;  // [0..1] live queue counters, [2..3] those of the last frame
    // tiled path
    int path = 0;                    // 0: tiled when the grid allows it, 1: always the direct path
    uint32_t pool_cap_option = 0;    // 0: sized from the triangle count
    int tile_shift_option = 0;       // 0: tile height chosen per frame; 3, 4, 5: fixed at 8, 16, 32 rows
    bool no_classes = false;         // AR_OPT_WORK_CLASSES 0: tile lists stay in entry order
    int ascii_path = 0;              // AR_OPT_SMOOTHING_KERNEL
    bool no_fold = false;            // AR_OPT_FOLD_SCAN 0: k_binscan in every frame
    bool blocksum_zero[kSets] = {};  // the totals behind the count table of a set are zero (left so by a frame without k_binscan)
    float *d_tripos = nullptr;       // 9 planes x tripos_stride floats: triangle positions
    size_t tripos_stride = 0;
    float *d_triattr = nullptr;      // n_tris x 24 floats: the three vertices' normals and colours
    // everything the binning kernels write, per buffer set
    TriStatic *d_tri = nullptr;      // n_tris 64-byte triangle records (positions, flat normal and colour, flags) in slot order
    TriStatic *d_tri_t = nullptr;    // the same records in the order of the triangles' own indices (shading)
    uint32_t *d_perm = nullptr;      // slot -> triangle index (the spatial order chosen at upload)
    float *d_box = nullptr;          // bounding box of every block of 256 slots
    void *d_rt = nullptr;            // ray backend: the scene arrays of ar_rt_set_scene in one allocation
    RtScene rt_scene = {};           // counts and device pointers into d_rt
    bool rt_has_scene = false;
    float *d_upd = nullptr;          // staging for ar_update_positions, grown on demand
    size_t upd_floats = 0;
    hipStream_t up_stream = nullptr; // its upload stream: the host waits for the copy into the staging buffer, never for frames
    hipEvent_t ev_upd[kSets] = {};   // ar_update_positions: frames in flight done / update done
    hipEvent_t ev_staged = nullptr, ev_taken = nullptr;  // staging buffer filled / consumed by the previous update
    bool upd_pending = false;        // ev_taken has been recorded
    uint4 *d_triuv = nullptr;        // n_tris texel-coordinate rows (ar_set_mesh_uv), else null
    uint32_t *d_atlas = nullptr;     // atlas texels (ar_set_texture), else null
    int atlas_w = 0, atlas_h = 0;
    uint2 *d_surv[kSets] = {};  // per k_setup workgroup: cell ranges of its surviving sub-triangles
    uint32_t *d_sid[kSets] = {};  // ... and their ids, 2*triangle + sub
    uint32_t *d_surv_count[kSets] = {};
    uint32_t *d_dircount[kSets] = {};  // per k_setup workgroup: entries per bin (+ scan scratch)
    size_t group_cap = 0, dircount_cap = 0;
    uint32_t *d_entries[kSets] = {};  // (2*triangle + sub, tile mask) entries grouped by bin
    uint32_t pool_cap = 0;
    BinState *d_state = nullptr;     // kSets blocks
    BatchSet batch[kSets];           // ar_render_batch_async
    uint32_t batch_no = 0;           // batches enqueued so far
    int last_batch = -1, last_batch_frames = 0, batch_cols = 0, batch_rows = 0;  // set, frame count and grid of the most recent batch
    uint32_t frame = 0;              // tiled frames enqueued so far
    bool last_tiled = false;
    uint8_t *d_q_rgba = nullptr, *d_q_ascii = nullptr, *d_q_plane = nullptr;  // ar_quantize staging: grid, codes, ramp-index plane
    size_t q_cells = 0;
    uint32_t *d_q_colors = nullptr;
    size_t qc_cells = 0;
    uint8_t *d_p_atlas = nullptr; float *d_p_cov = nullptr; uint32_t *d_p_image = nullptr;  // ar_present scratch
    size_t p_atlas_bytes = 0, p_pixels = 0;
    bool rendered = false;
    bool profiling = false;
    struct Stamp { hipEvent_t a, b; int slot; };
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> free_events;
    char err[256] = "";
};

namespace {

int fail(ar_ctx *ctx, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx ? ctx->err : g_create_error, 256, fmt, ap);
    va_end(ap);
    return code;
}

#define AR_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, AR_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void matmul_pv(const float *P, const float *V, float *PV) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            PV[c * 4 + r] = ((P[0 * 4 + r] * V[c * 4 + 0] + P[1 * 4 + r] * V[c * 4 + 1]) + P[2 * 4 + r] * V[c * 4 + 2]) +
                            P[3 * 4 + r] * V[c * 4 + 3];
}

struct Stamper {  // brackets one launch with events when profiling is on
    ar_ctx *ctx;
    hipEvent_t a = nullptr, b = nullptr;
    int slot;
    hipStream_t stream;
    Stamper(ar_ctx *c, int s, hipStream_t st = nullptr) : ctx(c), slot(s), stream(st ? st : c->stream) {
        if (!ctx->profiling) return;
        a = take(); b = take();
        if (a) (void)hipEventRecord(a, stream);
    }
    ~Stamper() {
        if (!ctx->profiling || !a || !b) return;
        (void)hipEventRecord(b, stream);
        ctx->stamps.push_back({a, b, slot});
    }
    hipEvent_t take() {
        if (!ctx->free_events.empty()) { hipEvent_t e = ctx->free_events.back(); ctx->free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
};

int sync_streams(ar_ctx *ctx) {
    ctx->overlapped_in_flight = false;
    for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipStreamSynchronize(ctx->fstream[k]));
    if (ctx->stream != ctx->own_stream) AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int free_targets(ar_ctx *ctx) {
    for (int k = 0; k < kSets; k++) {
        (void)hipFree(ctx->d_vis[k]); (void)hipFree(ctx->d_rgba[k]); (void)hipFree(ctx->d_depth[k]); (void)hipFree(ctx->d_ascii[k]);
        (void)hipFree(ctx->d_ridx[k]); (void)hipFree(ctx->d_ids[k]);
        ctx->d_vis[k] = nullptr; ctx->d_rgba[k] = nullptr; ctx->d_depth[k] = nullptr; ctx->d_ascii[k] = nullptr;
        ctx->d_ridx[k] = nullptr; ctx->d_ids[k] = nullptr;
    }
    ctx->cols = ctx->rows = 0;
    ctx->rendered = false;
    ctx->vis_clean = false;
    ctx->last_oset = 0;
    return 0;
}

int ensure_targets(ar_ctx *ctx, int cols, int rows) {
    const size_t n = size_t(cols) * rows;
    if (!(ctx->d_vis[0] && ctx->cols == cols && ctx->rows == rows)) {
        int rc = sync_streams(ctx);
        if (rc) return rc;
        free_targets(ctx);
        for (int k = 0; k < kSets; k++) {
            AR_HIP(ctx, hipMalloc(&ctx->d_vis[k], n * sizeof(uint64_t)));
            AR_HIP(ctx, hipMalloc(&ctx->d_rgba[k], n * 4));
            AR_HIP(ctx, hipMalloc(&ctx->d_depth[k], n * sizeof(float)));
            AR_HIP(ctx, hipMalloc(&ctx->d_ascii[k], n));
            AR_HIP(ctx, hipMalloc(&ctx->d_ridx[k], n));
            // rows outside a strip are never written by the kernels: give them defined contents once
            AR_HIP(ctx, hipMemsetAsync(ctx->d_rgba[k], 0, n * 4, ctx->stream));
            AR_HIP(ctx, hipMemsetAsync(ctx->d_depth[k], 0, n * sizeof(float), ctx->stream));
            AR_HIP(ctx, hipMemsetAsync(ctx->d_ascii[k], 0, n, ctx->stream));
            AR_HIP(ctx, hipMemsetAsync(ctx->d_ridx[k], 0, n, ctx->stream));
        }
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->cols = cols; ctx->rows = rows;
    }
    if (ctx->want_ids && !ctx->d_ids[0]) {
        int rc = sync_streams(ctx);
        if (rc) return rc;
        for (int k = 0; k < kSets; k++) {
            AR_HIP(ctx, hipMalloc(&ctx->d_ids[k], n * sizeof(uint32_t)));
            AR_HIP(ctx, hipMemsetAsync(ctx->d_ids[k], 0xFF, n * sizeof(uint32_t), ctx->stream));
        }
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

// Waits for the frames in flight and releases the current mesh with everything derived from it.
int drop_mesh(ar_ctx *ctx) {
    int rc = sync_streams(ctx);
    if (rc) return rc;
    (void)hipFree(ctx->d_vbo); (void)hipFree(ctx->d_idx); (void)hipFree(ctx->d_tripos); (void)hipFree(ctx->d_triattr);
    (void)hipFree(ctx->d_tri); (void)hipFree(ctx->d_triuv); (void)hipFree(ctx->d_tri_t); (void)hipFree(ctx->d_perm); (void)hipFree(ctx->d_box);
    ctx->d_vbo = nullptr; ctx->d_idx = nullptr; ctx->d_tripos = nullptr; ctx->d_triattr = nullptr;
    ctx->d_tri = nullptr; ctx->d_triuv = nullptr;  // texel coordinates belong to the mesh they were given for
    ctx->d_tri_t = nullptr; ctx->d_perm = nullptr; ctx->d_box = nullptr;
    ctx->n_verts = 0; ctx->n_tris = 0;
    return 0;
}

// d_vbo / d_idx hold the new mesh: choose the spatial order of its triangles (Morton code of the centroid, stable
// radix sort) and derive the streams of the tiled path in that order (position planes, triangle records, block
// boxes) plus the per-triangle copies the shading step reads (records, attribute rows).
int finish_mesh(ar_ctx *ctx, uint32_t n_verts, uint32_t n_tris) {
    ctx->tripos_stride = (size_t(n_tris) + 63) / 64 * 64;
    const size_t n_blocks = (size_t(n_tris) + 255) / 256;
    AR_HIP(ctx, hipMalloc(&ctx->d_tripos, 9 * ctx->tripos_stride * sizeof(float)));
    AR_HIP(ctx, hipMalloc(&ctx->d_triattr, size_t(n_tris) * 24 * sizeof(float)));
    AR_HIP(ctx, hipMalloc(&ctx->d_tri, size_t(n_tris) * sizeof(TriStatic)));
    AR_HIP(ctx, hipMalloc(&ctx->d_tri_t, size_t(n_tris) * sizeof(TriStatic)));
    AR_HIP(ctx, hipMalloc(&ctx->d_perm, size_t(n_tris) * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&ctx->d_box, n_blocks * 6 * sizeof(float)));
    uint32_t *d_code = nullptr, *d_code_sorted = nullptr, *d_order = nullptr, *d_bounds = nullptr;
    void *d_temp = nullptr;
    auto sorted = [&]() -> int {
        AR_HIP(ctx, hipMalloc(&d_code, size_t(n_tris) * sizeof(uint32_t)));
        AR_HIP(ctx, hipMalloc(&d_code_sorted, size_t(n_tris) * sizeof(uint32_t)));
        AR_HIP(ctx, hipMalloc(&d_order, size_t(n_tris) * sizeof(uint32_t)));
        AR_HIP(ctx, hipMalloc(&d_bounds, 6 * sizeof(uint32_t)));
        const uint32_t init[6] = {~0u, ~0u, ~0u, 0u, 0u, 0u};
        AR_HIP(ctx, hipMemcpyAsync(d_bounds, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (init lives on this stack frame)
        const unsigned blocks = unsigned(n_blocks);
        k_mesh_bounds<<<blocks < 2048u ? blocks : 2048u, 256, 0, ctx->stream>>>(ctx->d_vbo, ctx->d_idx, n_verts, n_tris, d_bounds);
        k_morton<<<blocks, 256, 0, ctx->stream>>>(ctx->d_vbo, ctx->d_idx, n_verts, n_tris, d_bounds, d_code, d_order);
        AR_HIP(ctx, hipGetLastError());
        size_t temp_bytes = 0;
        AR_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_code, d_code_sorted, d_order, ctx->d_perm, int(n_tris), 0, 30, ctx->stream));
        AR_HIP(ctx, hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
        AR_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_code, d_code_sorted, d_order, ctx->d_perm, int(n_tris), 0, 30, ctx->stream));
        k_tristream<true><<<blocks, 256, 0, ctx->stream>>>(ctx->d_vbo, ctx->d_idx, ctx->d_perm, n_verts, n_tris, ctx->tripos_stride, ctx->d_tripos,
                                                     ctx->d_triattr, ctx->d_tri, ctx->d_tri_t, ctx->d_box);
        AR_HIP(ctx, hipGetLastError());
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    };
    const int rc = sorted();
    (void)hipFree(d_code); (void)hipFree(d_code_sorted); (void)hipFree(d_order); (void)hipFree(d_bounds); (void)hipFree(d_temp);
    if (rc) return rc;
    ctx->n_verts = n_verts; ctx->n_tris = n_tris;
    return 0;
}

// The smoothing kernels store four codes at a time when the row length is a multiple of four, and bytes otherwise: a
// caller's ASCII target has to be 4-byte aligned in the first case only (row starts then are as well).
bool ascii_target_misaligned(const ar_ctx *ctx, int cols) {
    return ctx->ascii_target && (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(ctx->ascii_target) & 3u) != 0;
}

// What a frame of cols x rows cells looks like under the context's current state: fills fp (camera, strip, tile and bin
// grids, entry format - everything but the buffer set and the scan mode) and tells which path it takes.
struct FramePlan {
    bool tiled = false, textured = false;
    size_t nbins = 0, strip_cells = 0;
};

int plan_frame(ar_ctx *ctx, int cols, int rows, FramePlan &plan) {
    FrameParams &fp = ctx->fp;
    fp.cols = cols; fp.rows = rows;
    fp.hw = float(cols) * 0.5f; fp.hh = float(rows) * 0.5f;
    fp.n_tris = ctx->n_tris; fp.n_verts = ctx->n_verts;
    fp.want_ids = ctx->want_ids ? 1 : 0;
    int row0 = 0, row1 = rows;
    if (ctx->strip1 > ctx->strip0) {
        if (ctx->strip1 > rows) return fail(ctx, AR_E_ARG, "strip [%d,%d) exceeds %d rows", ctx->strip0, ctx->strip1, rows);
        row0 = ctx->strip0; row1 = ctx->strip1;
    }
    const int halo = fp.mode_on == 1 ? 3 : 0;  // smoothing reads up to 3 rows beyond the strip
    fp.row0 = row0; fp.row1 = row1;
    fp.ya = row0 - halo < 0 ? 0 : row0 - halo;
    fp.yb = row1 + halo > rows ? rows : row1 + halo;
    matmul_pv(ctx->proj, ctx->view, fp.pv);
    fp.has_model = ctx->has_model ? 1 : 0;
    if (ctx->has_model) {  // uProj * uView * uModel * v: GLSL multiplies the matrices first, left to right
        float pv[16];
        memcpy(pv, fp.pv, sizeof pv);
        matmul_pv(pv, ctx->model, fp.pv);
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) fp.model[3 * c + r] = ctx->model[4 * c + r];
    }
    // texturing needs texel coordinates, an atlas, and more than one texel either way (shader_utils.js:94)
    plan.textured = ctx->d_triuv && ctx->d_atlas && ctx->atlas_w > 1 && ctx->atlas_h > 1;
    fp.tex_w = plan.textured ? ctx->atlas_w : 0;
    fp.tex_h = plan.textured ? ctx->atlas_h : 0;

    plan.strip_cells = size_t(fp.yb - fp.ya) * cols;
    fp.ntx = (cols + kTileW - 1) / kTileW;
    // tile height: 32 rows, halved while the frame has fewer tiles than two per compute unit (strips, small grids)
    fp.tshift = kTileShiftMax;
    while (fp.tshift > kTileShiftMin && fp.ntx * (((fp.yb - fp.ya) + (1 << fp.tshift) - 1) >> fp.tshift) < 512) fp.tshift--;
    if (ctx->tile_shift_option) fp.tshift = ctx->tile_shift_option;
    fp.nty = ((fp.yb - fp.ya) + (1 << fp.tshift) - 1) >> fp.tshift;
    fp.bshift = 1;  // per-tile bins (0) shorten k_tile's filter but cost as much again in the binning kernels
    fp.nbx = (fp.ntx + (1 << fp.bshift) - 1) >> fp.bshift;
    fp.nby = (fp.nty + (1 << fp.bshift) - 1) >> fp.bshift;
    const size_t nt = size_t(fp.ntx) * fp.nty;
    const size_t nbins = size_t(fp.nbx) * fp.nby;
    plan.nbins = nbins;
    plan.tiled = ctx->path == 0 && nt <= size_t(kMaxTiles) && nbins <= size_t(kMaxBins) && cols <= 65535 &&
                       rows <= 65535 && ctx->d_tripos && ctx->n_tris <= (kEntryLocMask >> 1) - 65536u;  // slots fit an entry
    if (plan.tiled) {
        // one triangle per k_setup lane and pass; more passes only beyond kMaxGroups workgroups
        uint32_t chunk = (fp.n_tris + kMaxGroups - 1) / kMaxGroups;
        chunk = (chunk + kSetupThreads - 1) / kSetupThreads * kSetupThreads;
        fp.chunk = chunk;
        fp.n_groups = (fp.n_tris + chunk - 1) / chunk;
        fp.id_bits = 1;
        while (fp.id_bits < kEntryMaskShift && (2ull * fp.n_tris - 1) >> fp.id_bits) fp.id_bits++;
        fp.cls_bits = std::min(kEntryMaskShift - fp.id_bits, kMaxClassBits);
        if (ctx->no_classes) fp.cls_bits = 0;
        fp.bin_starts = 0;
        fp.fold_scan = (nbins <= size_t(kFoldScanBins) && !ctx->no_fold) ? 1 : 0;
    }
    return 0;
}

// Frames whose caller does not wait for them right away (may_overlap) run entirely on one of kSets frame
// streams, frame f on stream f % kSets, and every buffer a frame writes exists kSets times: consecutive
// frames share the GPU without a single cross-stream event (each of which costs the host about as much as
// two kernel launches), and frame f + kSets follows frame f on the same stream, which is all the ordering
// the reuse of a buffer set needs.
int enqueue_frame(ar_ctx *ctx, int cols, int rows, bool may_overlap) {
    if (cols <= 0 || rows <= 0 || size_t(cols) * rows > (size_t(1) << 31))
        return fail(ctx, AR_E_ARG, "ar_render: bad grid %d x %d", cols, rows);
    if (ascii_target_misaligned(ctx, cols))
        return fail(ctx, AR_E_ARG, "ar_render: the ASCII target must be 4-byte aligned when the row length (%d) is a multiple of four", cols);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_targets(ctx, cols, rows);
    if (rc) return rc;
    if (ctx->n_tris == 0 || !ctx->d_vbo || !ctx->d_idx) return 0;  // raster.js:274: nothing to draw

    FramePlan plan;
    rc = plan_frame(ctx, cols, rows, plan);
    if (rc) return rc;
    FrameParams &fp = ctx->fp;
    const bool tiled = plan.tiled, textured = plan.textured;
    const size_t nbins = plan.nbins;
    const unsigned cell_blocks = unsigned((plan.strip_cells + 255) / 256);
    const int set = int(ctx->frame % unsigned(kSets));  // buffer set of this frame (tiled path)
    const bool overlap = tiled && may_overlap && !ctx->ext_stream && ctx->overlap != 0;
    if (!overlap && ctx->overlapped_in_flight) {  // frames still running on the frame streams
        rc = sync_streams(ctx);
        if (rc) return rc;
    }
    if (overlap) ctx->overlapped_in_flight = true;
    hipStream_t fs = overlap ? ctx->fstream[set] : ctx->stream;  // the stream of this frame
    const int oset = overlap ? set : 0;                          // its output buffers
    uint8_t *ascii_out = ctx->ascii_target ? ctx->ascii_target - ptrdiff_t(ctx->target_row) * cols : ctx->d_ascii[oset];
    if (tiled != ctx->last_tiled && ctx->rendered) {
        // the two paths share the key buffers and do not order themselves against each other's streams
        rc = sync_streams(ctx);
        if (rc) return rc;
    }
    if (tiled) {
        fp.frame = ctx->frame;
        fp.set = set;
        const uint32_t want_cap = ctx->pool_cap_option ? ctx->pool_cap_option : 2u * fp.n_tris + 65536u;
        if (!ctx->d_entries[0] || ctx->pool_cap != want_cap) {
            rc = sync_streams(ctx);
            if (rc) return rc;
            for (int k = 0; k < kSets; k++) { (void)hipFree(ctx->d_entries[k]); ctx->d_entries[k] = nullptr; }
            ctx->pool_cap = 0;
            for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipMalloc(&ctx->d_entries[k], size_t(want_cap) * sizeof(uint32_t)));
            ctx->pool_cap = want_cap;
        }
        const size_t want_groups = size_t(fp.n_groups) * 2 * fp.chunk;
        if (!ctx->d_surv[0] || ctx->group_cap < want_groups) {
            rc = sync_streams(ctx);
            if (rc) return rc;
            for (int k = 0; k < kSets; k++) {
                (void)hipFree(ctx->d_surv[k]); (void)hipFree(ctx->d_sid[k]); (void)hipFree(ctx->d_surv_count[k]);
                ctx->d_surv[k] = nullptr; ctx->d_sid[k] = nullptr; ctx->d_surv_count[k] = nullptr;
            }
            ctx->group_cap = 0;
            for (int k = 0; k < kSets; k++) {
                AR_HIP(ctx, hipMalloc(&ctx->d_surv[k], want_groups * sizeof(uint2)));
                AR_HIP(ctx, hipMalloc(&ctx->d_sid[k], want_groups * sizeof(uint32_t)));
                AR_HIP(ctx, hipMalloc(&ctx->d_surv_count[k], (size_t(kMaxGroups) + 1) * sizeof(uint32_t)));
            }
            ctx->group_cap = want_groups;
        }
        const size_t want_dir = size_t(fp.n_groups) * nbins;
        if (!ctx->d_dircount[0] || ctx->dircount_cap < want_dir) {
            rc = sync_streams(ctx);
            if (rc) return rc;
            for (int k = 0; k < kSets; k++) { (void)hipFree(ctx->d_dircount[k]); ctx->d_dircount[k] = nullptr; ctx->blocksum_zero[k] = false; }
            ctx->dircount_cap = 0;
            for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipMalloc(&ctx->d_dircount[k], (want_dir + 16 * size_t(kMaxBins)) * sizeof(uint32_t)));
            ctx->dircount_cap = want_dir;
        }
    }
    const bool was_clean = ctx->vis_clean;
    ctx->vis_clean = false;  // stays false if anything below fails: the next frame then starts from scratch
    if (!was_clean) for (int k = 0; k < kSets; k++) ctx->blocksum_zero[k] = false;
    if (!was_clean) {
        // first frame on these targets (or after a failed one).  In steady state every kernel that
        // consumes a visibility key or a queue counter puts it back to its idle value, so no per-frame
        // clear is needed.
        rc = sync_streams(ctx);
        if (rc) return rc;
        AR_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(uint32_t), ctx->stream));
        AR_HIP(ctx, hipMemsetAsync(ctx->d_state, 0, kSets * sizeof(BinState), ctx->stream));
        {
            Stamper s(ctx, K_CLEAR);
            for (int k = 0; k < kSets; k++) k_clear<<<2048, 256, 0, ctx->stream>>>(ctx->d_vis[k], 0, size_t(cols) * rows);
        }
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (tiled) {
        uint32_t *dircount = ctx->d_dircount[set];
        uint32_t *blocksum = dircount + ctx->dircount_cap;
        // Frames with few bins (small grids, strips of a frame) do without k_binscan: one launch and one gap between
        // dependent kernels less, which is most of what such a frame costs.  They need the totals behind the count
        // table at zero; a frame of the other kind (or a failed one) leaves them otherwise.
        if (fp.fold_scan && !ctx->blocksum_zero[set])
            AR_HIP(ctx, hipMemsetAsync(blocksum, 0, 16 * size_t(kMaxBins) * sizeof(uint32_t), fs));
        ctx->blocksum_zero[set] = fp.fold_scan != 0;
        {
            Stamper s(ctx, K_SETUP, fs);
            k_setup<false><<<fp.n_groups, kSetupThreads, nbins * sizeof(uint32_t), fs>>>(fp, ctx->d_tripos, ctx->tripos_stride, ctx->d_box, ctx->d_surv[set], ctx->d_sid[set],
                                                         ctx->d_surv_count[set], dircount, blocksum, nullptr, BatchStrides{});
        }
        if (!fp.fold_scan) {
            Stamper s(ctx, K_BINSCAN, fs);
            k_binscan<false><<<dim3(unsigned((nbins + 63) / 64), 16), 64 * kScanWaves, 0, fs>>>(fp, dircount, blocksum, nullptr, BatchStrides{});
        }
        {
            Stamper s(ctx, K_FILL, fs);
            k_fill<false><<<fp.n_groups, kSetupThreads, nbins * 2 * sizeof(uint32_t) + ((nbins + 3) & ~size_t(3)), fs>>>(fp, ctx->d_tri, ctx->d_surv[set], ctx->d_sid[set], ctx->d_surv_count[set], dircount, blocksum,
                                                        ctx->d_state, ctx->d_entries[set], ctx->pool_cap, ctx->d_vis[set], nullptr, BatchStrides{});
        }
        {
            Stamper s(ctx, K_TILE, fs);
            const unsigned grid = unsigned((nbins + 7) / 8 * 8) << (2 * fp.bshift);
            auto kernel = fp.tshift == 5 ? (textured ? k_tile<true, 5, false> : k_tile<false, 5, false>)
                          : fp.tshift == 4 ? (textured ? k_tile<true, 4, false> : k_tile<false, 4, false>)
                                           : (textured ? k_tile<true, 3, false> : k_tile<false, 3, false>);
            kernel<<<grid, kTileThreads, 0, fs>>>(fp, ctx->d_vbo, ctx->d_idx, ctx->d_triattr, ctx->d_triuv,
                                         ctx->d_atlas, ctx->d_tri, ctx->d_tri_t, ctx->d_entries[set], ctx->pool_cap, ctx->d_state, blocksum,
                                         ctx->d_vis[set], ctx->d_rgba[oset], ctx->d_depth[oset], ascii_out, ctx->d_ridx[set],
                                         ctx->d_ids[oset], nullptr, BatchStrides{});
        }
        ctx->frame++;
    } else {
        {
            Stamper s(ctx, K_RASTER);
            k_raster<<<(fp.n_tris + 255) / 256, 256, 0, ctx->stream>>>(fp, ctx->d_vbo, ctx->d_idx, ctx->d_vis[0], ctx->d_medium,
                                                                      ctx->d_large, ctx->d_counters);
        }
        {
            Stamper s(ctx, K_MEDIUM);
            k_raster_medium<<<512, 256, 0, ctx->stream>>>(fp, ctx->d_vis[0], ctx->d_medium, ctx->d_counters);
        }
        {
            Stamper s(ctx, K_LARGE);
            dim3 grid((cols + 15) / 16, (fp.yb - fp.ya + 15) / 16);
            k_raster_large<<<grid, 256, 0, ctx->stream>>>(fp, ctx->d_vis[0], ctx->d_large, ctx->d_counters);
        }
        {
            Stamper s(ctx, K_SHADE);
            k_shade<<<cell_blocks, 256, 0, ctx->stream>>>(fp, ctx->d_vbo, ctx->d_idx, ctx->d_triuv, ctx->d_atlas, ctx->d_vis[0],
                                                         ctx->d_rgba[0], ctx->d_depth[0],
                                                         ascii_out, ctx->d_ridx[0], ctx->d_ids[0], ctx->d_counters);
        }
    }
    if (fp.mode_on == 1) {
        Stamper s(ctx, K_ASCII, fs);
        // a textured frame can hold ASCII overrides in its alpha bytes, which the plane of ramp indices cannot express
        if (textured) launch_ascii_rgba(fp, reinterpret_cast<const uint32_t *>(ctx->d_rgba[oset]), ctx->d_ridx[tiled ? set : 0], false, ascii_out, fs);
        else launch_ascii<false>(fp, ctx->d_ridx[tiled ? set : 0], ascii_out, fs);
    }
    AR_HIP(ctx, hipGetLastError());
    ctx->vis_clean = true;
    ctx->rendered = true;
    ctx->last_tiled = tiled;
    ctx->last_oset = oset;
    ctx->last_stream = fs;
    ctx->last_slot = overlap ? set : kSets;
    return 0;
}

void free_batch_set(BatchSet &b) {
    (void)hipFree(b.d_surv); (void)hipFree(b.d_sid); (void)hipFree(b.d_surv_count); (void)hipFree(b.d_dircount); (void)hipFree(b.d_entries);
    (void)hipFree(b.d_state); (void)hipFree(b.d_vis); (void)hipFree(b.d_ridx); (void)hipFree(b.d_rgba); (void)hipFree(b.d_ascii);
    (void)hipFree(b.d_depth); (void)hipFree(b.d_ids); (void)hipFree(b.d_fps);
    if (b.h_fps) (void)hipHostFree(b.h_fps);
    if (b.ev_params) (void)hipEventDestroy(b.ev_params);
    b = BatchSet();
}

// Makes batch set k hold n frames of the planned frame in ctx->fp (grows, never shrinks, while the grid stays the same).
int ensure_batch_set(ar_ctx *ctx, int k, int n, int cols, int rows, const FramePlan &plan) {
    BatchSet &b = ctx->batch[k];
    const FrameParams &fp = ctx->fp;
    const size_t cells = size_t(cols) * rows;
    const size_t surv = size_t(fp.n_groups) * 2 * fp.chunk, table = size_t(fp.n_groups) * plan.nbins, dir = table + 16 * size_t(kMaxBins);
    const size_t entries = ctx->pool_cap_option ? ctx->pool_cap_option : 2u * size_t(fp.n_tris) + 65536u;
    if (b.frames >= n && b.cells == cells && b.surv >= surv && b.dir_table == table && b.entries == entries && (b.ids || !ctx->want_ids)) return 0;
    int rc = sync_streams(ctx);
    if (rc) return rc;
    free_batch_set(b);
    const size_t N = size_t(n);
    AR_HIP(ctx, hipMalloc(&b.d_surv, N * surv * sizeof(uint2)));
    AR_HIP(ctx, hipMalloc(&b.d_sid, N * surv * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&b.d_surv_count, N * (size_t(kMaxGroups) + 1) * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&b.d_dircount, N * dir * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&b.d_entries, N * entries * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&b.d_state, N * sizeof(BinState)));
    AR_HIP(ctx, hipMalloc(&b.d_vis, N * cells * sizeof(uint64_t)));
    AR_HIP(ctx, hipMalloc(&b.d_ridx, N * cells));
    AR_HIP(ctx, hipMalloc(&b.d_rgba, N * cells * 4));
    AR_HIP(ctx, hipMalloc(&b.d_ascii, N * cells));
    AR_HIP(ctx, hipMalloc(&b.d_depth, N * cells * sizeof(float)));
    if (ctx->want_ids) AR_HIP(ctx, hipMalloc(&b.d_ids, N * cells * sizeof(uint32_t)));
    AR_HIP(ctx, hipMalloc(&b.d_fps, N * sizeof(FrameParams)));
    AR_HIP(ctx, hipHostMalloc(&b.h_fps, N * sizeof(FrameParams), hipHostMallocDefault));
    AR_HIP(ctx, hipEventCreateWithFlags(&b.ev_params, hipEventDisableTiming));
    // defined contents for what the kernels do not write (rows outside a strip), idle values for what they expect to find idle
    hipStream_t s = ctx->own_stream;
    AR_HIP(ctx, hipMemsetAsync(b.d_state, 0, N * sizeof(BinState), s));
    AR_HIP(ctx, hipMemsetAsync(b.d_dircount, 0, N * dir * sizeof(uint32_t), s));
    AR_HIP(ctx, hipMemsetAsync(b.d_ridx, 0, N * cells, s));
    AR_HIP(ctx, hipMemsetAsync(b.d_rgba, 0, N * cells * 4, s));
    AR_HIP(ctx, hipMemsetAsync(b.d_ascii, 0, N * cells, s));
    AR_HIP(ctx, hipMemsetAsync(b.d_depth, 0, N * cells * sizeof(float), s));
    if (b.d_ids) AR_HIP(ctx, hipMemsetAsync(b.d_ids, 0xFF, N * cells * sizeof(uint32_t), s));
    k_clear<<<2048, 256, 0, s>>>(b.d_vis, 0, N * cells);
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipStreamSynchronize(s));
    b.blocksum_zero = true;
    b.frames = n; b.cells = cells; b.surv = surv; b.dircount = dir; b.dir_table = table; b.entries = entries; b.ids = ctx->want_ids;
    return 0;
}

// n frames, cameras[f] = (proj, view) of frame f, in ONE chain of launches: every kernel of the tiled path runs once
// for all frames (frame = blockIdx.z).  A frame that is bound by launch and hand-over latencies rather than by work - a
// small scene, a strip of a frame on one of several GPUs - costs a fraction of a launch chain this way.  Frames that
// cannot take that road (the direct path, textured frames) are rendered one by one into the same buffers.
int enqueue_batch(ar_ctx *ctx, int cols, int rows, int n, const float *cameras, size_t ascii_stride) {
    if (cols <= 0 || rows <= 0 || size_t(cols) * rows > (size_t(1) << 31)) return fail(ctx, AR_E_ARG, "ar_render_batch: bad grid %d x %d", cols, rows);
    if (n <= 0 || n > kMaxBatchFrames || !cameras) return fail(ctx, AR_E_ARG, "ar_render_batch: 1..%d frames and their cameras expected (got %d)", kMaxBatchFrames, n);
    const size_t cells = size_t(cols) * rows;
    if (ctx->ascii_target && n > 1 && ascii_stride < size_t(cols))
        return fail(ctx, AR_E_ARG, "ar_render_batch: with an ASCII target the frames' planes must lie at least a row apart (stride %zu)", ascii_stride);
    if (ascii_target_misaligned(ctx, cols) || (ctx->ascii_target && (cols & 3) == 0 && (ascii_stride & 3u) != 0))
        return fail(ctx, AR_E_ARG, "ar_render_batch: the ASCII target and its stride must be 4-byte aligned when the row length (%d) is a multiple of four", cols);
    AR_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n_tris == 0 || !ctx->d_vbo || !ctx->d_idx) return 0;  // raster.js:274: nothing to draw
    float keep_proj[16], keep_view[16];
    memcpy(keep_proj, ctx->proj, sizeof keep_proj);
    memcpy(keep_view, ctx->view, sizeof keep_view);
    auto restore = [&]() { memcpy(ctx->proj, keep_proj, sizeof keep_proj); memcpy(ctx->view, keep_view, sizeof keep_view); };
    FramePlan plan;
    int rc = plan_frame(ctx, cols, rows, plan);
    if (rc) return rc;
    const int k = int(ctx->batch_no % unsigned(kSets));
    rc = ensure_batch_set(ctx, k, n, cols, rows, plan);
    if (rc) return rc;
    BatchSet &b = ctx->batch[k];
    const bool overlap = plan.tiled && !plan.textured && !ctx->ext_stream && ctx->overlap != 0;
    if (!overlap && ctx->overlapped_in_flight) {
        rc = sync_streams(ctx);
        if (rc) return rc;
    }
    hipStream_t fs = overlap ? ctx->fstream[k] : ctx->stream;
    uint8_t *ascii_base = ctx->ascii_target ? ctx->ascii_target - ptrdiff_t(ctx->target_row) * cols : b.d_ascii;
    const size_t ascii_step = ctx->ascii_target ? ascii_stride : cells;

    if (!plan.tiled || plan.textured) {
        // one by one through the single-frame path (its own buffers), each copied into the batch's planes behind it
        uint8_t *keep_target = ctx->ascii_target;
        for (int f = 0; f < n && !rc; f++) {
            memcpy(ctx->proj, cameras + size_t(f) * 32, 16 * sizeof(float));
            memcpy(ctx->view, cameras + size_t(f) * 32 + 16, 16 * sizeof(float));
            if (keep_target) ctx->ascii_target = keep_target + size_t(f) * ascii_stride;
            rc = enqueue_frame(ctx, cols, rows, false);
            if (rc) break;
            hipStream_t s = ctx->stream;
            hipError_t e = hipMemcpyAsync(b.d_rgba + size_t(f) * cells * 4, ctx->d_rgba[0], cells * 4, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(b.d_depth + size_t(f) * cells, ctx->d_depth[0], cells * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && !keep_target) e = hipMemcpyAsync(b.d_ascii + size_t(f) * cells, ctx->d_ascii[0], cells, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && ctx->want_ids && b.d_ids && ctx->d_ids[0])
                e = hipMemcpyAsync(b.d_ids + size_t(f) * cells, ctx->d_ids[0], cells * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) rc = fail(ctx, AR_E_HIP, "ar_render_batch: copy failed: %s", hipGetErrorString(e));
        }
        ctx->ascii_target = keep_target;
        restore();
        if (rc) return rc;
        ctx->batch_no++;
        ctx->last_batch = k; ctx->last_batch_frames = n; ctx->batch_cols = cols; ctx->batch_rows = rows;
        ctx->last_stream = ctx->stream; ctx->last_slot = kSets;
        return 0;
    }

    if (overlap) ctx->overlapped_in_flight = true;
    // the frames' parameters: the planned frame with each frame's camera, staged in page-locked memory
    if (b.params_pending) AR_HIP(ctx, hipEventSynchronize(b.ev_params));  // (the copy of the batch that used this set last: long done)
    for (int f = 0; f < n; f++) {
        memcpy(ctx->proj, cameras + size_t(f) * 32, 16 * sizeof(float));
        memcpy(ctx->view, cameras + size_t(f) * 32 + 16, 16 * sizeof(float));
        FramePlan again;
        rc = plan_frame(ctx, cols, rows, again);
        if (rc) { restore(); return rc; }
        b.h_fps[f] = ctx->fp;
        // (leaving the scan kernel out saves a frame on its own a launch, at the price of more work in k_fill; a chain
        // of several frames pays one launch for all of them, so there the scan kernel stays)
        if (n >= 4) { b.h_fps[f].fold_scan = 0; b.h_fps[f].bin_starts = 1; }
        b.h_fps[f].set = f;       // index of the frame's binning state in the batch's array
        b.h_fps[f].frame = uint32_t(f);
    }
    restore();
    const FrameParams &fp = b.h_fps[0];
    AR_HIP(ctx, hipMemcpyAsync(b.d_fps, b.h_fps, size_t(n) * sizeof(FrameParams), hipMemcpyHostToDevice, fs));
    AR_HIP(ctx, hipEventRecord(b.ev_params, fs));
    b.params_pending = true;
    const size_t nbins = plan.nbins;
    BatchStrides bs;
    bs.surv = b.surv; bs.groups = size_t(kMaxGroups) + 1; bs.dircount = b.dircount; bs.entries = b.entries; bs.cells = cells; bs.ascii = ascii_step; bs.frames = uint32_t(n);
    uint32_t *dircount = b.d_dircount, *blocksum = b.d_dircount + b.dir_table;
    const unsigned N = unsigned(n);
    // (frames without k_binscan want the totals behind the count table at zero: k_tile leaves them so, the allocation made them so)
    if (fp.fold_scan && !b.blocksum_zero) AR_HIP(ctx, hipMemsetAsync(b.d_dircount, 0, size_t(b.frames) * b.dircount * sizeof(uint32_t), fs));
    b.blocksum_zero = fp.fold_scan != 0;  // (frames without k_binscan put the totals back to zero, the others leave k_binscan's there)
    {
        Stamper s(ctx, K_SETUP, fs);
        k_setup<true><<<dim3((fp.n_groups + 7u) / 8u * 8u * N), kSetupThreads, nbins * sizeof(uint32_t), fs>>>(fp, ctx->d_tripos, ctx->tripos_stride, ctx->d_box, b.d_surv, b.d_sid,
                                                                                              b.d_surv_count, dircount, blocksum, b.d_fps, bs);
    }
    if (!fp.fold_scan) {
        Stamper s(ctx, K_BINSCAN, fs);
        k_binscan<true><<<dim3(unsigned((nbins + 63) / 64), 16, N), 64 * kScanWaves, 0, fs>>>(fp, dircount, blocksum, b.d_fps, bs);
        if (fp.bin_starts) k_binstart<true><<<dim3(1, 1, N), kSetupThreads, 0, fs>>>(fp, blocksum, b.d_state, uint32_t(b.entries), b.d_fps, bs);
    }
    {
        Stamper s(ctx, K_FILL, fs);
        k_fill<true><<<dim3(fp.n_groups, 1, N), kSetupThreads, nbins * 2 * sizeof(uint32_t) + ((nbins + 3) & ~size_t(3)), fs>>>(
            fp, ctx->d_tri, b.d_surv, b.d_sid, b.d_surv_count, dircount, blocksum, b.d_state, b.d_entries, uint32_t(b.entries), b.d_vis, b.d_fps, bs);
    }
    {
        Stamper s(ctx, K_TILE, fs);
        const unsigned grid = unsigned((nbins + 7) / 8 * 8) << (2 * fp.bshift);
        auto kernel = fp.tshift == 5 ? k_tile<false, 5, true> : fp.tshift == 4 ? k_tile<false, 4, true> : k_tile<false, 3, true>;
        kernel<<<dim3(grid * N), kTileThreads, 0, fs>>>(fp, ctx->d_vbo, ctx->d_idx, ctx->d_triattr, ctx->d_triuv, ctx->d_atlas, ctx->d_tri, ctx->d_tri_t,
                                                         b.d_entries, uint32_t(b.entries), b.d_state, blocksum, b.d_vis, b.d_rgba, b.d_depth, ascii_base,
                                                         b.d_ridx, b.d_ids, b.d_fps, bs);
    }
    if (fp.mode_on == 1) {
        Stamper s(ctx, K_ASCII, fs);
        launch_ascii_planes(fp, b.d_ridx, ascii_base, fs, n, b.d_fps, bs);
    }
    AR_HIP(ctx, hipGetLastError());
    ctx->batch_no++;
    ctx->last_batch = k; ctx->last_batch_frames = n; ctx->batch_cols = cols; ctx->batch_rows = rows;
    ctx->last_stream = fs; ctx->last_slot = overlap ? k : kSets;
    return 0;
}

}  // namespace
