// raster_direct.h - direct path: k_clear, k_raster, k_raster_medium, k_raster_large, k_shade (global 64-bit atomic min)
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

__global__ __launch_bounds__(256) void k_clear(uint64_t *__restrict__ vis, size_t first, size_t count) {
    const size_t n = size_t(blockIdx.x) * 256 + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t p = n; p < count; p += stride) vis[first + p] = kBackgroundKey;
}

// counters: [0] medium records, [1] large records
__global__ __launch_bounds__(256) void k_raster(FrameParams fp, const float *__restrict__ vbo,
                                                const uint32_t *__restrict__ idx, uint64_t *__restrict__ vis,
                                                TriRec *__restrict__ medium, TriRec *__restrict__ large,
                                                uint32_t *__restrict__ counters) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= fp.n_tris) return;
    const uint32_t v0 = idx[3 * size_t(t)], v1 = idx[3 * size_t(t) + 1], v2 = idx[3 * size_t(t) + 2];
    if (v0 >= fp.n_verts || v1 >= fp.n_verts || v2 >= fp.n_verts) return;
    const Vert<0> in0 = load_vertex<0>(fp, vbo, v0), in1 = load_vertex<0>(fp, vbo, v1), in2 = load_vertex<0>(fp, vbo, v2);
    Vert<0> s0[3], s1[3];
    const int ns = clip_near<0>(in0, in1, in2, s0, s1);
    const int jlo = fp.rows - fp.yb, jhi = fp.rows - 1 - fp.ya;  // window rows of the rendered strip
    for (int s = 0; s < ns; s++) {
        const Vert<0> t0 = s ? s1[0] : s0[0], t1 = s ? s1[1] : s0[1], t2 = s ? s1[2] : s0[2];
        Screen sc;
        if (!to_screen(t0.c, t1.c, t2.c, fp.hw, fp.hh, sc) || !front_facing(sc)) continue;
        int i0, i1, j0, j1;
        if (!cell_range(sc.X, sc.Y, sc.W, fp.cols, jlo, jhi, i0, i1, j0, j1)) continue;
        Edges e;
        edges_from_screen(sc, e);
        const uint32_t id = 2u * t + uint32_t(s);
        const int ncells = (i1 - i0 + 1) * (j1 - j0 + 1);
        if (ncells > kSmallMaxCells) {
            const bool is_large = ncells > kMediumMaxCells;
            const uint32_t cap = is_large ? kLargeCapacity : kMediumCapacity;
            const uint32_t slot = atomicAdd(&counters[is_large ? 1 : 0], 1u);
            if (slot < cap) {
                TriRec r;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    r.a[k] = e.a[k]; r.b[k] = e.b[k]; r.c[k] = e.c[k];
                    r.W[k] = e.W[k]; r.Z[k] = e.Z[k];
                }
                r.i0 = i0; r.i1 = i1; r.j0 = j0; r.j1 = j1;
                r.id = id; r.pad = 0;
                (is_large ? large : medium)[slot] = r;
                continue;
            }
            // queue full: fall through and scan the range here (slow but correct)
        }
        for (int j = j0; j <= j1; j++) {
            const size_t row = size_t(fp.rows - 1 - j) * fp.cols;
            for (int i = i0; i <= i1; i++) {
                float f[3], zw;
                if (cover(e, i, j, f, zw)) vis_min(vis, row + i, make_key(zw, id));
            }
        }
    }
}

// One wavefront per queued record; each step covers an 8x8 block of the record's cell range.
__global__ __launch_bounds__(256) void k_raster_medium(FrameParams fp, uint64_t *__restrict__ vis,
                                                       const TriRec *__restrict__ medium,
                                                       const uint32_t *__restrict__ counters) {
    const uint32_t count = min(counters[0], kMediumCapacity);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4u;
    const int lx = int(lane & 7u), ly = int(lane >> 3);
    for (uint32_t r = wave; r < count; r += n_waves) {
        const TriRec &rec = medium[r];
        const int i0 = rec.i0, i1 = rec.i1, j0 = rec.j0, j1 = rec.j1;
        for (int by = j0; by <= j1; by += 8) {
            const int j = by + ly;
            for (int bx = i0; bx <= i1; bx += 8) {
                const int i = bx + lx;
                float f[3], zw;
                if (i <= i1 && j <= j1 && cover(rec, i, j, f, zw))
                    vis_min(vis, size_t(fp.rows - 1 - j) * fp.cols + i, make_key(zw, rec.id));
            }
        }
    }
}

// One lane per cell of a 16x16 tile; every tile walks the list of large records.
__global__ __launch_bounds__(256) void k_raster_large(FrameParams fp, uint64_t *__restrict__ vis,
                                                      const TriRec *__restrict__ large,
                                                      const uint32_t *__restrict__ counters) {
    const uint32_t count = min(counters[1], kLargeCapacity);
    if (count == 0) return;
    const int tx0 = int(blockIdx.x) * 16, ty0 = fp.ya + int(blockIdx.y) * 16;  // output (top-down) rows
    const int i = tx0 + int(threadIdx.x & 15u), r = ty0 + int(threadIdx.x >> 4);
    const int j = fp.rows - 1 - r;
    const int tj1 = fp.rows - 1 - ty0, tj0 = tj1 - 15;  // the tile's window-row span
    const bool live = (i < fp.cols) && (r < fp.yb);
    uint64_t best = kBackgroundKey;
    for (uint32_t k = 0; k < count; k++) {
        const TriRec &rec = large[k];
        if (rec.i1 < tx0 || rec.i0 > tx0 + 15 || rec.j1 < tj0 || rec.j0 > tj1) continue;  // wave-uniform
        float f[3], zw;
        if (live && i >= rec.i0 && i <= rec.i1 && j >= rec.j0 && j <= rec.j1 && cover(rec, i, j, f, zw)) {
            const uint64_t key = make_key(zw, rec.id);
            best = key < best ? key : best;
        }
    }
    if (live && best != kBackgroundKey) vis_min(vis, size_t(r) * fp.cols + i, best);
}

// Colour and depth of cell (i, j) (GL window coordinates) for winner id = 2*triangle + sub: rebuild the
// (sub-)triangle from the mesh with the varyings, evaluate the same edge expressions as the raster
// pass (which reproduces its depth bit for bit), interpolate perspective-correctly and shade.
// triuv: per triangle (u0 | v0 << 16, u1 | v1 << 16, u2 | v2 << 16, flags); read only while texturing is on.
__device__ __forceinline__ void shade_cell(const FrameParams &fp, const float *__restrict__ vbo,
                                           const uint32_t *__restrict__ idx, const uint4 *__restrict__ triuv,
                                           const uint32_t *__restrict__ atlas, uint32_t id, int i, int j,
                                           uint8_t (&px)[4], float &zw) {
    const uint32_t t = id >> 1, sub = id & 1u;
    const uint32_t v0 = idx[3 * size_t(t)], v1 = idx[3 * size_t(t) + 1], v2 = idx[3 * size_t(t) + 2];
    Vert<11> in0 = load_vertex<11>(fp, vbo, v0), in1 = load_vertex<11>(fp, vbo, v1), in2 = load_vertex<11>(fp, vbo, v2);
    bool textured = false;
    if (fp.tex_w > 0) {
        const uint4 uv = triuv[t];
        textured = (uv.w & kUvSample) != 0u;
        if (textured) {
            in0.v[9] = float(uv.x & 0xFFFFu); in0.v[10] = float(uv.x >> 16);
            in1.v[9] = float(uv.y & 0xFFFFu); in1.v[10] = float(uv.y >> 16);
            in2.v[9] = float(uv.z & 0xFFFFu); in2.v[10] = float(uv.z >> 16);
        }
    }
    Vert<11> s0[3], s1[3];
    clip_near<11>(in0, in1, in2, s0, s1);
    const Vert<11> t0 = sub ? s1[0] : s0[0], t1 = sub ? s1[1] : s0[1], t2 = sub ? s1[2] : s0[2];
    Screen sc;
    to_screen(t0.c, t1.c, t2.c, fp.hw, fp.hh, sc);
    Edges e;
    edges_from_screen(sc, e);
    float f[3];
    cover(e, i, j, f, zw);
    const float rS = 1.0f / ((f[0] + f[1]) + f[2]);
    const float b0 = f[0] * rS, b1 = f[1] * rS, b2 = f[2] * rS;
    float v[11];
#pragma unroll
    for (int q = 0; q < 11; q++) v[q] = (b0 * t0.v[q] + b1 * t1.v[q]) + b2 * t2.v[q];
    if (textured && apply_texture(fp, atlas, v[9], v[10], v + 6, px)) return;
    shade(fp, v, px);
}

// One lane per cell of output rows [ya, yb): rebuild the winner's setup, interpolate, shade, store.
// The cell's visibility key is put back to "background" here, so the next frame needs no clear pass;
// lane 0 of block 0 retires the frame's queue counters (cnt[0..1] live, cnt[2..3] = last frame).
__global__ __launch_bounds__(256) void k_shade(FrameParams fp, const float *__restrict__ vbo,
                                               const uint32_t *__restrict__ idx, const uint4 *__restrict__ triuv,
                                               const uint32_t *__restrict__ atlas, uint64_t *__restrict__ vis,
                                               uint8_t *__restrict__ rgba, float *__restrict__ depth,
                                               uint8_t *__restrict__ ascii, uint8_t *__restrict__ ridx,
                                               uint32_t *__restrict__ ids, uint32_t *__restrict__ cnt) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt[2] = cnt[0]; cnt[3] = cnt[1];
        cnt[0] = 0; cnt[1] = 0;
    }
    const size_t n = size_t(fp.yb - fp.ya) * fp.cols;
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const size_t cell = size_t(fp.ya) * fp.cols + p;
    const int r = int(cell / fp.cols), i = int(cell - size_t(r) * fp.cols);
    const int j = fp.rows - 1 - r;
    const uint64_t key = vis[cell];
    uint8_t px[4] = {0, 0, 0, 255};
    float zw = 1.0f;
    if (key != kBackgroundKey) {
        vis[cell] = kBackgroundKey;
        shade_cell(fp, vbo, idx, triuv, atlas, uint32_t(key), i, j, px, zw);
    }
    *reinterpret_cast<uint32_t *>(rgba + cell * 4) =
        uint32_t(px[0]) | (uint32_t(px[1]) << 8) | (uint32_t(px[2]) << 16) | (uint32_t(px[3]) << 24);
    depth[cell] = zw;
    if (fp.want_ids) ids[cell] = uint32_t(key);
    const int ri = ramp_index(px[0], px[1], px[2], fp.ramp_len);
    if (fp.mode_on) ridx[cell] = uint8_t(ri);  // input of the smoothing pass
    else if (r >= fp.row0 && r < fp.row1) ascii[cell] = (px[3] >= 2 && px[3] <= 254) ? px[3] : fp.ramp[ri];  // override: textured cells only
}

}  // namespace
