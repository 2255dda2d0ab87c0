// ascii_rast.hip - MI355X (gfx950) 3D -> ASCII rasterizer: HIP kernels and the C ABI around them.
//
// Two resolve paths produce identical frames (nothing leaves HBM between their stages):
//   tiled (default)   k_setup -> k_binscan -> k_binoffsets -> k_fill -> k_tile -> k_ascii_plane
//                     triangles are set up once, binned to screen bins and resolved per 64x32-cell
//                     tile in LDS; described in front of k_tristream.
//   direct            k_raster -> k_raster_medium -> k_raster_large -> k_shade -> k_ascii_plane
//                     k_raster: one lane per triangle (vertex transform, near clip, edge setup, cull,
//                     cell range); small triangles are scanned in place, larger ones queued as setup
//                     records for k_raster_medium (one wavefront per record, 8x8 cells per step) and
//                     k_raster_large (one lane per cell scanning the short list of screen-filling
//                     records); k_shade: one lane per cell (winner's varyings, Lambert + point lights,
//                     RGBA8, depth, ramp quantisation; ASCII written here when smoothing is off).
//                     Used for grids beyond the tiled path's tile budget, or on request.
// Visibility is resolved with one 64-bit key per cell, (depth bits << 32) | order, reduced with unsigned
// min: smaller depth wins, equal depth goes to the earlier triangle (order = 2*triangle + sub on the
// direct path, the record slot - handed out in triangle order - on the tiled path), and the result
// does not depend on the order in which lanes, waves or kernels reach a cell.
//
// Arithmetic follows a fixed contract (binary32 transforms and shading, binary64 edge functions,
// no fused multiply-add, correctly rounded division and square root) so that the output is
// reproducible bit for bit; build with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "ascii_rast.h"

namespace {

constexpr uint64_t kBackgroundKey = (uint64_t(0x3F800000u) << 32) | 0xFFFFFFFFull;  // depth 1.0f
constexpr int kSmallMaxCells = 64;      // cell-range area a single lane scans itself
constexpr int kMediumMaxCells = 4096;   // above this a record goes to the per-cell pass
constexpr uint32_t kMediumCapacity = 1u << 18;
constexpr uint32_t kLargeCapacity = 1u << 14;

enum KernelSlot { K_CLEAR = 0, K_RASTER, K_MEDIUM, K_LARGE, K_SHADE, K_ASCII, K_SETUP, K_BINSCAN, K_BINOFF, K_FILL, K_TILE, K_PRESENT, K_RAYTRACE };
const char *const kKernelNames[AR_N_KERNELS] = {"k_clear",  "k_raster", "k_raster_medium", "k_raster_large",
                                                "k_shade", "k_ascii",  "k_setup",         "k_binscan",
                                                "k_binoffsets", "k_fill", "k_tile", "k_present",
                                                "k_raytrace"};

// Everything a frame's kernels need besides the big buffers; passed by value as a kernel argument.
struct FrameParams {
    float pv[16];  // P*V, column-major
    float amb[3], ldir[3], lcol[3];
    float plpos[3 * AR_MAX_POINT_LIGHTS], plcol[3 * AR_MAX_POINT_LIGHTS];
    int n_pl;
    int cols, rows;
    int ya, yb;      // output rows [ya, yb) that are rasterised and shaded (strip + smoothing halo)
    int row0, row1;  // output rows [row0, row1) whose ASCII codes are written
    float hw, hh;    // cols/2, rows/2
    uint32_t n_tris, n_verts;
    int ramp_len, mode_on, mode_radius, mode_thresh;
    int want_ids;    // also store the winning (2*triangle + sub) per cell
    int ntx, nty;    // tile grid over columns [0, cols) x output rows [ya, yb)
    int nbx, nby;    // bin grid
    int bshift;      // a bin is (1 << bshift)^2 tiles: 0 while the tiles fit the bin table, else 1
    uint32_t chunk, n_groups;  // triangles per k_setup workgroup, number of those workgroups
    uint32_t frame;  // frame counter
    int set;         // buffer set of this frame (frame % kSets): binning state block, record slots, ...
    int tex_w, tex_h;  // atlas size in texels; 0 x 0 while texturing is off
    uint8_t ramp[AR_MAX_RAMP];
};

// Setup record of one (sub-)triangle: edge functions, per-vertex w and clip z, cell range, key id.
struct TriRec {
    double a[3], b[3], c[3];
    float W[3], Z[3];
    int i0, i1, j0, j1;  // inclusive cell range, GL window rows (y up)
    uint32_t id;         // 2*triangle + sub-triangle
    uint32_t pad;
};

template <int NV>
struct Vert {
    float c[4];                // clip x, y, z, w
    float v[NV > 0 ? NV : 1];  // varyings (world position, normal, colour) when NV == 9
};

// gl_Position = uProj * uView * vec4(a_pos, 1) with PV precomputed on the host.
template <int NV>
__device__ __forceinline__ Vert<NV> load_vertex(const FrameParams &fp, const float *__restrict__ vbo, uint32_t vi) {
    const float *p = vbo + size_t(vi) * 9;
    Vert<NV> o;
    const float x = p[0], y = p[1], z = p[2];
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.c[r] = ((fp.pv[0 * 4 + r] * x + fp.pv[1 * 4 + r] * y) + fp.pv[2 * 4 + r] * z) + fp.pv[3 * 4 + r];
    if (NV > 0) {
        o.v[0] = x; o.v[1] = y; o.v[2] = z;
#pragma unroll
        for (int k = 3; k < NV; k++) o.v[k] = k < 9 ? p[k] : 0.0f;  // beyond the buffer's nine: texel coordinates, set by the caller
    }
    return o;
}

// New vertex on the near plane, always from the inside vertex I towards the outside vertex O, so
// two triangles sharing the edge compute the identical point.
template <int NV>
__device__ __forceinline__ Vert<NV> clip_vertex(const Vert<NV> &I, float dI, const Vert<NV> &O, float dO) {
    Vert<NV> r;
    const float t = dI / (dI - dO);
#pragma unroll
    for (int k = 0; k < 4; k++) r.c[k] = I.c[k] + (O.c[k] - I.c[k]) * t;
#pragma unroll
    for (int k = 0; k < NV; k++) r.v[k] = I.v[k] + (O.v[k] - I.v[k]) * t;
    return r;
}

// Near-plane clip (inside: z_c + w_c >= 0).  Returns the number of sub-triangles (0..2) in s0 / s1.
template <int NV>
__device__ __forceinline__ int clip_near(const Vert<NV> &in0, const Vert<NV> &in1, const Vert<NV> &in2,
                                         Vert<NV> (&s0)[3], Vert<NV> (&s1)[3]) {
    const float d0 = in0.c[2] + in0.c[3], d1 = in1.c[2] + in1.c[3], d2 = in2.c[2] + in2.c[3];
    const bool b0 = d0 >= 0.0f, b1 = d1 >= 0.0f, b2 = d2 >= 0.0f;
    const int n_in = int(b0) + int(b1) + int(b2);
    if (n_in == 0) return 0;
    if (n_in == 3) {
        s0[0] = in0; s0[1] = in1; s0[2] = in2;
        return 1;
    }
    // rotate so that A is the odd vertex out: the only inside one, or the only outside one
    const int a = (n_in == 1) ? (b0 ? 0 : (b1 ? 1 : 2)) : (!b0 ? 0 : (!b1 ? 1 : 2));
    Vert<NV> A, B, C;
    float dA, dB, dC;
    if (a == 0) { A = in0; B = in1; C = in2; dA = d0; dB = d1; dC = d2; }
    else if (a == 1) { A = in1; B = in2; C = in0; dA = d1; dB = d2; dC = d0; }
    else { A = in2; B = in0; C = in1; dA = d2; dB = d0; dC = d1; }
    if (n_in == 1) {
        s0[0] = A;
        s0[1] = clip_vertex<NV>(A, dA, B, dB);
        s0[2] = clip_vertex<NV>(A, dA, C, dC);
        return 1;
    }
    const Vert<NV> pab = clip_vertex<NV>(B, dB, A, dA);
    const Vert<NV> pca = clip_vertex<NV>(C, dC, A, dA);
    s0[0] = pab; s0[1] = B; s0[2] = C;
    s1[0] = pab; s1[1] = C; s1[2] = pca;
    return 2;
}

struct Edges {
    double a[3], b[3], c[3];
    float W[3], Z[3];
};

// Pixel-homogeneous vertices of one (sub-)triangle: (X, Y, W) = ((x_c + w_c) cols/2, (y_c + w_c) rows/2, w_c),
// GL window orientation (y up), plus clip z for the depth interpolation.
struct Screen {
    float X[3], Y[3], W[3], Z[3];
};

// False when the triangle cannot produce fragments because a vertex is non-finite or has w <= 0.
__device__ __forceinline__ bool to_screen(const float (&c0)[4], const float (&c1)[4], const float (&c2)[4], float hw,
                                          float hh, Screen &s) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 4; q++) ok = ok && isfinite(c0[q]) && isfinite(c1[q]) && isfinite(c2[q]);
    ok = ok && (c0[3] > 0.0f) && (c1[3] > 0.0f) && (c2[3] > 0.0f);
    if (!ok) return false;
    s.W[0] = c0[3]; s.W[1] = c1[3]; s.W[2] = c2[3];
    s.Z[0] = c0[2]; s.Z[1] = c1[2]; s.Z[2] = c2[2];
    s.X[0] = (c0[0] + c0[3]) * hw; s.X[1] = (c1[0] + c1[3]) * hw; s.X[2] = (c2[0] + c2[3]) * hw;
    s.Y[0] = (c0[1] + c0[3]) * hh; s.Y[1] = (c1[1] + c1[3]) * hh; s.Y[2] = (c2[1] + c2[3]) * hh;
    return true;
}

// Edge functions e_k = v_{k+1} x v_{k+2} in binary64 (every product of two binary32 values is exact).
template <typename S>
__device__ __forceinline__ void edges_from_screen(const S &s, Edges &e) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        e.a[k] = double(s.Y[p]) * double(s.W[q]) - double(s.Y[q]) * double(s.W[p]);
        e.b[k] = double(s.X[q]) * double(s.W[p]) - double(s.X[p]) * double(s.W[q]);
        e.c[k] = double(s.X[p]) * double(s.Y[q]) - double(s.X[q]) * double(s.Y[p]);
        e.W[k] = s.W[k];
        e.Z[k] = s.Z[k];
    }
}

// det = v0 . (v1 x v2) > 0: counter-clockwise in the window, i.e. front facing; <= 0 is culled
// (back face or zero area).
__device__ __forceinline__ bool front_facing(const Screen &s) {
    const double a0 = double(s.Y[1]) * double(s.W[2]) - double(s.Y[2]) * double(s.W[1]);
    const double b0 = double(s.X[2]) * double(s.W[1]) - double(s.X[1]) * double(s.W[2]);
    const double c0 = double(s.X[1]) * double(s.Y[2]) - double(s.X[2]) * double(s.Y[1]);
    const double det = (double(s.X[0]) * a0 + double(s.Y[0]) * b0) + double(s.W[0]) * c0;
    return det > 0.0;
}

// Inclusive cell range (GL window coordinates) that contains every covered cell centre, clamped to
// the viewport columns and to window rows [jlo, jhi].  False when empty.
__device__ __forceinline__ bool cell_range(const float (&X)[3], const float (&Y)[3], const float (&W)[3], int cols,
                                           int jlo, int jhi, int &i0, int &i1, int &j0, int &j1) {
    float minx = X[0] / W[0], maxx = minx, miny = Y[0] / W[0], maxy = miny;
#pragma unroll
    for (int k = 1; k < 3; k++) {
        const float sx = X[k] / W[k], sy = Y[k] / W[k];
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx);
        miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    const float lox = ceilf(fmaxf(minx - 0.50390625f, 0.0f)), hix = floorf(fminf(maxx - 0.49609375f, float(cols - 1)));
    const float loy = ceilf(fmaxf(miny - 0.50390625f, float(jlo))), hiy = floorf(fminf(maxy - 0.49609375f, float(jhi)));
    if (!(lox <= hix) || !(loy <= hiy)) return false;
    i0 = int(lox); i1 = int(hix); j0 = int(loy); j1 = int(hiy);
    return true;
}

__device__ __forceinline__ bool edge_inside(double E, double a, double b) {
    return E > 0.0 || (E == 0.0 && (a > 0.0 || (a == 0.0 && b > 0.0)));
}

// Coverage test of cell (i, j) (GL window coordinates) against one record; on a hit f receives the
// binary32 edge values and zw the window depth.  Depth outside [0, 1) is no hit.
template <typename R>
__device__ __forceinline__ bool cover(const R &e, int i, int j, float (&f)[3], float &zw) {
    const double px = double(i) + 0.5, py = double(j) + 0.5;
    const double E0 = e.a[0] * px + (e.b[0] * py + e.c[0]);  // row term first: it is invariant along a row
    const double E1 = e.a[1] * px + (e.b[1] * py + e.c[1]);
    const double E2 = e.a[2] * px + (e.b[2] * py + e.c[2]);
    if (!edge_inside(E0, e.a[0], e.b[0]) || !edge_inside(E1, e.a[1], e.b[1]) || !edge_inside(E2, e.a[2], e.b[2]))
        return false;
    f[0] = float(E0); f[1] = float(E1); f[2] = float(E2);
    const float den = (f[0] * e.W[0] + f[1] * e.W[1]) + f[2] * e.W[2];
    const float num = (f[0] * e.Z[0] + f[1] * e.Z[1]) + f[2] * e.Z[2];
    const float zn = num / den;
    zw = zn * 0.5f + 0.5f;
    return (zw >= 0.0f) && (zw < 1.0f);
}

__device__ __forceinline__ uint64_t make_key(float zw, uint32_t id) {
    return (uint64_t(__float_as_uint(zw)) << 32) | uint64_t(id);
}

__device__ __forceinline__ void vis_min(uint64_t *vis, size_t cell, uint64_t key) {
    // the plain load only filters: a stale (larger) value costs one redundant atomic, never a wrong result
    if (key < vis[cell]) atomicMin(reinterpret_cast<unsigned long long *>(vis + cell), (unsigned long long)key);
}

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

__device__ __forceinline__ int ramp_index(int r, int g, int b, int ramp_len) {
    // == clamp(round(((r+g+b)/3/255) * (L-1)), 0, L-1): the rounded value is never a tie
    return (2 * (r + g + b) * (ramp_len - 1) + 765) / 1530;
}

// raster_shader.js:42-62 in binary32.
__device__ __forceinline__ void shade(const FrameParams &fp, const float *v, uint8_t (&out)[4]) {
    const float len2 = (v[3] * v[3] + v[4] * v[4]) + v[5] * v[5];
    const float inv = (len2 > 0.0f) ? 1.0f / sqrtf(len2) : 0.0f;
    const float N[3] = {v[3] * inv, v[4] * inv, v[5] * inv};
    float col[3];
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = v[6 + k] * fp.amb[k];
    const float ndl = fmaxf((N[0] * (-fp.ldir[0]) + N[1] * (-fp.ldir[1])) + N[2] * (-fp.ldir[2]), 0.0f);
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] = col[k] + (v[6 + k] * fp.lcol[k]) * ndl;
    for (int i = 0; i < AR_MAX_POINT_LIGHTS; i++) {
        if (i >= fp.n_pl) break;
        const float lv[3] = {fp.plpos[3 * i] - v[0], fp.plpos[3 * i + 1] - v[1], fp.plpos[3 * i + 2] - v[2]};
        const float d2 = fmaxf((lv[0] * lv[0] + lv[1] * lv[1]) + lv[2] * lv[2], 1e-4f);
        const float il = 1.0f / sqrtf(d2);
        const float Ld[3] = {lv[0] * il, lv[1] * il, lv[2] * il};
        const float ndlp = fmaxf((N[0] * Ld[0] + N[1] * Ld[1]) + N[2] * Ld[2], 0.0f);
        const float atten = 1.0f / (1.0f + d2 * 0.05f);
        const float w = ndlp * atten;
#pragma unroll
        for (int k = 0; k < 3; k++) col[k] = col[k] + (v[6 + k] * fp.plcol[3 * i + k]) * w;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = fminf(fmaxf(col[k], 0.0f), 1.0f);
        out[k] = uint8_t(int(floorf(c * 255.0f + 0.5f)));
    }
    out[3] = 255;
}

// Atlas lookup of one fragment at interpolated texel coordinates (u, v) (shader_utils.js:108-117: nearest
// texel floor(uv + 0.5), nothing outside the atlas; fetchAtlas :100-105 reads row height-1-y of the
// uploaded array).  Texel classes as the reference's primary ray treats them (pathtrace_shader.js:141-152):
// alpha 32..126 - the texel's RGB goes out unshaded with that alpha, an ASCII override for the
// quantizer (returns true, px is final); alpha 1 - the texel's rgb / 255 replaces the colour varying;
// anything else - untextured.
__device__ __forceinline__ bool apply_texture(const FrameParams &fp, const uint32_t *__restrict__ atlas, float u, float v,
                                              float *col, uint8_t (&px)[4]) {
    const float tx = floorf(u + 0.5f), ty = floorf(v + 0.5f);
    if (!(tx >= 0.0f) || !(ty >= 0.0f) || !(tx < float(fp.tex_w)) || !(ty < float(fp.tex_h))) return false;
    const uint32_t q = atlas[size_t(fp.tex_h - 1 - int(ty)) * fp.tex_w + int(tx)];
    const uint32_t a = q >> 24;
    if (a >= 32u && a <= 126u) {
        px[0] = uint8_t(q); px[1] = uint8_t(q >> 8); px[2] = uint8_t(q >> 16); px[3] = uint8_t(a);
        return true;
    }
    if (a == 1u) {
        col[0] = float(q & 255u) / 255.0f; col[1] = float((q >> 8) & 255u) / 255.0f; col[2] = float((q >> 16) & 255u) / 255.0f;
    }
    return false;
}

constexpr uint32_t kUvSample = 1u;  // flags word of a triangle's texel-coordinate row: sample the atlas

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

// ------------------------------------------------------------------------------------------------
// Tiled path: sub-triangles are binned to screen bins of 2x2 tiles, each 64x32-cell tile is resolved
// in LDS.
//
//   k_tristream (at mesh upload) gathers, per triangle, the three positions into nine planes of
//             n_tris floats (36 contiguous, coalesced bytes per triangle for the per-frame pass),
//             the three vertices' normals and colours into one 96-byte row (one gather per shaded
//             cell instead of index -> vertex chasing) and a flag telling whether the triangle is
//             flat (identical normal and colour at its three vertices).
//   k_setup   one pass over the positions, one triangle per lane and pass: transform, near clip,
//             cull, cell range.  Every sub-triangle that can produce fragments gets a 64-byte setup
//             record in the workgroup's block of record slots (dense, in triangle order), a 16-byte
//             tuple for k_fill, and is counted into the workgroup's LDS histogram over bins, which
//             is then published as one row of the count table.
//   k_binscan column-wise exclusive prefix of the count table over the k_setup workgroups (in
//             place, two levels) and per-row-block bin totals: every (workgroup, bin) pair gets a
//             fixed slice of the bin's entry array without a single global atomic.
//   k_binoffsets  bin totals, start of each bin's entry array (exclusive scan), row-block offsets,
//             overflow flag.
//   k_fill    every k_setup workgroup's survivors become one entry (record slot, mask of the bin's
//             tiles the cell range touches) per (survivor, bin), written into that workgroup's slice.
//   k_tile    one workgroup per tile: filters its bin's entries by mask bit into an LDS work list,
//             rasterises those records into a 2048-entry LDS key tile with LDS atomic min (small
//             cell ranges: one lane per record; large ones: all lanes sweep the range), then shades
//             the covered cells and writes the tile's RGBA / depth / ramp-index rows.  Visibility
//             keys never touch HBM.
// If the entry pool cannot hold a bin, the sub-triangles touching that bin are rasterised straight
// into a global key buffer by k_fill, k_binoffsets raises the frame's overflow flag and k_tile merges
// that buffer in.
// Everything a frame writes exists kSets times, so that consecutive frames can overlap: see
// enqueue_frame.
// ------------------------------------------------------------------------------------------------
constexpr int kSets = 3;               // frames in flight: every per-frame buffer exists kSets times
constexpr int kTileShiftH = 5;          // log2 of the tile height
constexpr int kTileW = 64, kTileH = 1 << kTileShiftH, kTileCells = kTileW * kTileH;  // (a tile is one wavefront wide)
constexpr int kMaxTiles = 8000;       // tiles per frame the tiled path accepts
constexpr int kMaxBins = 2048;        // >= kMaxTiles / 4 for any tile-grid shape the path accepts
constexpr int kSetupThreads = 256;    // k_setup / k_fill workgroup size
constexpr uint32_t kMaxGroups = 1024;  // k_setup workgroups (rows of the count table)
constexpr int kTileBigMax = 512;      // per-tile list of records swept cooperatively
constexpr int kTileSmallCells = 64;   // cell-range area (inside the tile) one lane scans itself
constexpr int kTileListMax = 2048;    // ids staged in LDS between the filter and the raster step
constexpr int kEntryMaskShift = 28;   // a bin entry is (record slot) | (mask of the bin's 2x2 tiles) << 28
constexpr uint32_t kEntryLocMask = (1u << kEntryMaskShift) - 1u;
constexpr int kFilterBatch = 4;       // bin entries a k_tile lane loads per filter step
constexpr uint32_t kFlagClipped = 1u; // record produced by the near-plane clip (vertices differ from the mesh's)
constexpr uint32_t kFlagFlat = 2u;    // the triangle's three vertices carry identical normal and colour
constexpr uint32_t kFlagTextured = 4u;  // the triangle samples the atlas (set from its texel-coordinate row)

struct __align__(16) BinRec {
    float X[3], Y[3], W[3], Z[3];
    uint32_t id;              // 2*triangle + sub-triangle
    uint16_t i0, i1, j0, j1;  // inclusive cell range, GL window rows (y up)
    uint32_t flags;           // kFlagClipped | kFlagFlat | kFlagTextured
};
static_assert(sizeof(BinRec) == 64, "bin records are four 16-byte loads");

// Per-frame binning state; one block per buffer set.
struct BinState {
    uint32_t total[kMaxBins];
    uint32_t start[kMaxBins];   // exclusive scan of total (k_binoffsets)
    uint32_t overflow;          // some bin did not fit the entry pool
    uint32_t n_entries;         // sum of total
    uint32_t pad[2];
};

// Exclusive prefix sum of v over the lanes of the workgroup (lane order); total = sum of all.
// NW = wavefronts per workgroup.
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(incl, d);
        if (lane >= uint32_t(d)) incl += n;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t x = s_wave[w];
        if (uint32_t(w) < wave) off += x;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return off + incl - v;
}

// Per triangle: positions -> planes pos[k*stride + t] (k = 3*vertex + axis); normal and colour of each
// vertex -> attr[t*24 + 8*v ..] = (nx, ny, nz, r, g, b, 0, 0), so a vertex is one 16-byte plus one
// 8-byte load; flat[t] = kFlagFlat when the three normals and the three colours are identical.
// A triangle that names a vertex outside the buffer gets NaN positions, which every later stage rejects.
__global__ __launch_bounds__(256) void k_tristream(const float *__restrict__ vbo, const uint32_t *__restrict__ idx,
                                                   uint32_t n_verts, uint32_t n_tris, size_t stride,
                                                   float *__restrict__ pos, float *__restrict__ attr,
                                                   uint8_t *__restrict__ flat) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tris) return;
    float a[3][9];
    bool all_ok = true;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const uint32_t vi = idx[3 * size_t(t) + v];
        const bool ok = vi < n_verts;
        all_ok = all_ok && ok;
        const float *p = vbo + size_t(ok ? vi : 0u) * 9;
#pragma unroll
        for (int q = 0; q < 9; q++) a[v][q] = ok ? p[q] : 0.0f;
#pragma unroll
        for (int ax = 0; ax < 3; ax++) pos[size_t(3 * v + ax) * stride + t] = ok ? p[ax] : __builtin_nanf("");
    }
    float4 *row = reinterpret_cast<float4 *>(attr) + size_t(t) * 6;
    bool is_flat = all_ok;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        row[2 * v] = make_float4(a[v][3], a[v][4], a[v][5], a[v][6]);
        row[2 * v + 1] = make_float4(a[v][7], a[v][8], 0.0f, 0.0f);
#pragma unroll
        for (int q = 3; q < 9; q++) is_flat = is_flat && (__float_as_uint(a[v][q]) == __float_as_uint(a[0][q]));
    }
    flat[t] = is_flat ? uint8_t(kFlagFlat) : uint8_t(0);
}

// Indexed mesh -> the reference's unshared, flat-shaded vertex buffer (what _rebuildGeometry's pushTri emits
// per triangle, raster.js:158-169,222-236): one lane per triangle gathers its three positions, takes the
// normal normalize(cross(b - a, c - a)) in binary64 - every product and sum rounded on its own, as the
// JavaScript expression is - with (0, 1, 0) when its length is <= 1e-8, and writes three vertices of
// (position, normal, colour) plus the indices 3t, 3t+1, 3t+2.  A vertex index beyond the position array
// yields NaN positions (what a typed array yields there), which every later stage rejects.
__global__ __launch_bounds__(256) void k_build_vbo(const float *__restrict__ positions, uint32_t n_verts,
                                                   const uint32_t *__restrict__ indices, uint32_t n_tris,
                                                   const float *__restrict__ tri_colors, float *__restrict__ vbo,
                                                   uint32_t *__restrict__ idx) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tris) return;
    float p[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t vi = indices[3 * size_t(t) + k];
#pragma unroll
        for (int ax = 0; ax < 3; ax++) p[k][ax] = vi < n_verts ? positions[3 * size_t(vi) + ax] : __builtin_nanf("");
    }
    const double ux = double(p[1][0]) - double(p[0][0]), uy = double(p[1][1]) - double(p[0][1]), uz = double(p[1][2]) - double(p[0][2]);
    const double vx = double(p[2][0]) - double(p[0][0]), vy = double(p[2][1]) - double(p[0][1]), vz = double(p[2][2]) - double(p[0][2]);
    double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const double L = sqrt((nx * nx + ny * ny) + nz * nz);
    if (L > 1e-8) { nx /= L; ny /= L; nz /= L; } else { nx = 0.0; ny = 1.0; nz = 0.0; }
    const float r = tri_colors ? tri_colors[3 * size_t(t)] : 0.8f, g = tri_colors ? tri_colors[3 * size_t(t) + 1] : 0.8f,
                b = tri_colors ? tri_colors[3 * size_t(t) + 2] : 0.8f;  // raster.js:145: default albedo
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float *o = vbo + (3 * size_t(t) + k) * 9;
        o[0] = p[k][0]; o[1] = p[k][1]; o[2] = p[k][2];
        o[3] = float(nx); o[4] = float(ny); o[5] = float(nz);
        o[6] = r; o[7] = g; o[8] = b;
        idx[3 * size_t(t) + k] = 3u * t + uint32_t(k);
    }
}

// Copies the "sample the atlas" bit of every triangle's texel-coordinate row into its flag byte (all bits
// cleared when triuv is null).
__global__ __launch_bounds__(256) void k_uvflags(const uint4 *__restrict__ triuv, uint32_t n_tris, uint8_t *__restrict__ flat) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tris) return;
    const bool sample = triuv && (triuv[t].w & kUvSample) != 0u;
    flat[t] = uint8_t((flat[t] & ~kFlagTextured) | (sample ? kFlagTextured : 0u));
}

__device__ __forceinline__ Vert<0> transform_position(const FrameParams &fp, float x, float y, float z) {
    Vert<0> o;
#pragma unroll
    for (int r = 0; r < 4; r++)
        o.c[r] = ((fp.pv[0 * 4 + r] * x + fp.pv[1 * 4 + r] * y) + fp.pv[2 * 4 + r] * z) + fp.pv[3 * 4 + r];
    return o;
}

// Range of bins a cell range touches.
__device__ __forceinline__ void bin_range(const FrameParams &fp, int i0, int i1, int j0, int j1, int &bx0, int &bx1,
                                          int &by0, int &by1) {
    bx0 = i0 >> (6 + fp.bshift); bx1 = i1 >> (6 + fp.bshift);
    by0 = (fp.rows - 1 - j1 - fp.ya) >> (kTileShiftH + fp.bshift); by1 = (fp.rows - 1 - j0 - fp.ya) >> (kTileShiftH + fp.bshift);
}

// One sub-triangle that survived setup: screen-space vertices and inclusive cell range.
struct Candidate {
    Screen sc;
    int i0, i1, j0, j1;
};

// Setup of the triangle with vertex positions p[0..8]: transform, near clip, cull, cell range within
// window rows [jlo, jhi].  c[s] / have[s] describe sub-triangle s; clipped tells whether the near
// plane cut the triangle.
__device__ __forceinline__ void triangle_setup(const FrameParams &fp, const float (&p)[9], int jlo, int jhi,
                                               Candidate (&c)[2], bool (&have)[2], bool &clipped) {
    const Vert<0> in0 = transform_position(fp, p[0], p[1], p[2]), in1 = transform_position(fp, p[3], p[4], p[5]),
                  in2 = transform_position(fp, p[6], p[7], p[8]);
    Vert<0> s0[3], s1[3];
    const int ns = clip_near<0>(in0, in1, in2, s0, s1);
    clipped = ns == 2 || (in0.c[2] + in0.c[3] < 0.0f) || (in1.c[2] + in1.c[3] < 0.0f) || (in2.c[2] + in2.c[3] < 0.0f);
    have[0] = have[1] = false;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (s >= ns) continue;
        const Vert<0> t0 = s ? s1[0] : s0[0], t1 = s ? s1[1] : s0[1], t2 = s ? s1[2] : s0[2];
        if (!to_screen(t0.c, t1.c, t2.c, fp.hw, fp.hh, c[s].sc) || !front_facing(c[s].sc)) continue;
        have[s] = cell_range(c[s].sc.X, c[s].sc.Y, c[s].sc.W, fp.cols, jlo, jhi, c[s].i0, c[s].i1, c[s].j0, c[s].j1);
    }
}

__device__ __forceinline__ BinRec make_record(const Candidate &c, uint32_t id, uint32_t flags) {
    BinRec r;
#pragma unroll
    for (int q = 0; q < 3; q++) { r.X[q] = c.sc.X[q]; r.Y[q] = c.sc.Y[q]; r.W[q] = c.sc.W[q]; r.Z[q] = c.sc.Z[q]; }
    r.id = id;
    r.i0 = uint16_t(c.i0); r.i1 = uint16_t(c.i1); r.j0 = uint16_t(c.j0); r.j1 = uint16_t(c.j1);
    r.flags = flags;
    return r;
}

// Workgroup g owns record slots [g * 2 * chunk, (g + 1) * 2 * chunk) and fills them densely, in
// triangle order: a record's slot number ("loc") therefore grows with its id, and visibility keys
// carry loc in place of the id - same tie-break, and the winner's record is found without a lookup.
// surv: the cell ranges of the same slots, (i0 | i1 << 16, j0 | j1 << 16), all k_fill needs of a record.
// One triangle per lane and pass; chunk is a multiple of the workgroup size (one pass up to about
// four million triangles).
__global__ __launch_bounds__(kSetupThreads) void k_setup(FrameParams fp, const float *__restrict__ tripos, size_t stride,
                                                         const uint8_t *__restrict__ flat, BinRec *__restrict__ records,
                                                         uint2 *__restrict__ surv, uint32_t *__restrict__ surv_count,
                                                         uint32_t *__restrict__ dircount) {
    __shared__ uint32_t hist[kMaxBins];
    __shared__ uint32_t s_wave[kSetupThreads / 64];
    const uint32_t tid = threadIdx.x;
    const int nbins = fp.nbx * fp.nby;
    const uint32_t t_begin = blockIdx.x * fp.chunk;
    const uint32_t t_end = min(t_begin + fp.chunk, fp.n_tris);
    for (int k = int(tid); k < nbins; k += kSetupThreads) hist[k] = 0;
    __syncthreads();
    const int jlo = fp.rows - fp.yb, jhi = fp.rows - 1 - fp.ya;
    const uint32_t loc_base = blockIdx.x * 2u * fp.chunk;
    uint32_t run = 0;  // slots handed out so far (identical in every lane)

    for (uint32_t tb = t_begin; tb < t_end; tb += kSetupThreads) {
        const uint32_t t = tb + tid;
        const bool live = t < t_end;
        Candidate c[2];
        bool have[2] = {false, false}, clipped = false;
        uint32_t flags = 0;
        if (live) {
            float p[9];
#pragma unroll
            for (int q = 0; q < 9; q++) p[q] = tripos[size_t(q) * stride + t];
            flags = flat[t];  // kFlagFlat | kFlagTextured
            triangle_setup(fp, p, jlo, jhi, c, have, clipped);
            if (clipped) flags |= kFlagClipped;
        }
        uint32_t total;
        uint32_t slot = run + block_exclusive_scan<kSetupThreads / 64>(uint32_t(have[0]) + uint32_t(have[1]), s_wave, total);
        run += total;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (!have[s]) continue;
            const uint32_t loc = loc_base + slot;
            slot++;
            records[loc] = make_record(c[s], 2u * t + uint32_t(s), flags);
            surv[loc] = make_uint2(uint32_t(c[s].i0) | (uint32_t(c[s].i1) << 16), uint32_t(c[s].j0) | (uint32_t(c[s].j1) << 16));
            int bx0, bx1, by0, by1;
            bin_range(fp, c[s].i0, c[s].i1, c[s].j0, c[s].j1, bx0, bx1, by0, by1);
            for (int by = by0; by <= by1; by++)
                for (int bx = bx0; bx <= bx1; bx++) atomicAdd(&hist[by * fp.nbx + bx], 1u);
        }
    }
    __syncthreads();
    for (int k = int(tid); k < nbins; k += kSetupThreads) dircount[size_t(blockIdx.x) * nbins + k] = hist[k];
    if (tid == 0) surv_count[blockIdx.x] = run;
}

// Column-wise exclusive prefix of the count table over the k_setup workgroups, in two levels.  The
// table's rows are cut into 16 row blocks; workgroup (x, y) takes bins [64x, 64x + 64) of row block y
// (its 16 wavefronts split the block's rows), rewrites table[g][b] as "entries of bin b that belong to
// earlier workgroups of the same row block" and stores the block's total in blocksum[y][b].  k_fill
// adds the totals of the earlier row blocks.  No global atomics, and the layout of every bin's entry
// array is the same on every run.
__global__ __launch_bounds__(1024) void k_binscan(FrameParams fp, uint32_t *__restrict__ table, uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t part[16][64];
    const int nbins = fp.nbx * fp.nby;
    const int lane = int(threadIdx.x & 63u), w = int(threadIdx.x >> 6);
    const int b = int(blockIdx.x) * 64 + lane;
    const uint32_t rows_per_block = (fp.n_groups + 15u) / 16u;
    const uint32_t blk0 = min(uint32_t(blockIdx.y) * rows_per_block, fp.n_groups), blk1 = min(blk0 + rows_per_block, fp.n_groups);
    const uint32_t rows_per_wave = (rows_per_block + 15u) / 16u;  // <= 16 (at most 4096 k_setup workgroups)
    const uint32_t g0 = min(blk0 + uint32_t(w) * rows_per_wave, blk1), g1 = min(g0 + rows_per_wave, blk1);
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k] = (b < nbins && g0 + k < g1) ? table[size_t(g0 + k) * nbins + b] : 0u;
        sum += v[k];
    }
    part[w][lane] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t x = part[k][lane];
        if (k < w) run += x;
        tot += x;
    }
    if (b < nbins) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (g0 + k < g1) table[size_t(g0 + k) * nbins + b] = run;
            run += v[k];
        }
        if (w == 0) blocksum[size_t(blockIdx.y) * nbins + b] = tot;
    }
}

// One workgroup: per bin, exclusive prefix of the 16 row-block totals and the bin's total; an
// exclusive scan of the totals gives the start of each bin's entry array.  boff[y][b] = start of
// bin b + entries of row blocks before y, so that k_fill only adds its own scanned table entry.
__global__ __launch_bounds__(1024) void k_binoffsets(FrameParams fp, const uint32_t *__restrict__ blocksum,
                                                     uint32_t *__restrict__ boff, BinState *__restrict__ state,
                                                     uint32_t entries_cap) {
    __shared__ uint32_t s_wave[16];
    const int nbins = fp.nbx * fp.nby;
    const uint32_t tid = threadIdx.x;
    BinState &st = state[fp.set];
    const int b0 = 2 * int(tid), b1 = b0 + 1;  // kMaxBins == 2 * 1024
    uint32_t x0[16], x1[16], t0 = 0, t1 = 0;
#pragma unroll
    for (int y = 0; y < 16; y++) {
        x0[y] = b0 < nbins ? blocksum[size_t(y) * nbins + b0] : 0u;
        x1[y] = b1 < nbins ? blocksum[size_t(y) * nbins + b1] : 0u;
        t0 += x0[y]; t1 += x1[y];
    }
    uint32_t n_entries;
    const uint32_t s0 = block_exclusive_scan<16>(t0 + t1, s_wave, n_entries), s1 = s0 + t0;
    const bool ok0 = s0 <= entries_cap && t0 <= entries_cap - s0, ok1 = s1 <= entries_cap && t1 <= entries_cap - s1;
    uint32_t r0 = s0, r1 = s1;
#pragma unroll
    for (int y = 0; y < 16; y++) {
        if (b0 < nbins) boff[size_t(y) * nbins + b0] = r0;
        if (b1 < nbins) boff[size_t(y) * nbins + b1] = r1;
        r0 += x0[y]; r1 += x1[y];
    }
    if (b0 < nbins) { st.start[b0] = s0; st.total[b0] = t0; }
    if (b1 < nbins) { st.start[b1] = s1; st.total[b1] = t1; }
    const int any_overflow = __syncthreads_or((b0 < nbins && !ok0 && t0) || (b1 < nbins && !ok1 && t1));
    if (tid == 0) { st.n_entries = n_entries; st.overflow = any_overflow ? 1u : 0u; }
}

__global__ __launch_bounds__(kSetupThreads) void k_fill(FrameParams fp, const BinRec *__restrict__ records,
                                                        const uint2 *__restrict__ surv,
                                                        const uint32_t *__restrict__ surv_count,
                                                        const uint32_t *__restrict__ dircount,
                                                        const uint32_t *__restrict__ boff, const BinState *__restrict__ state,
                                                        uint32_t *__restrict__ entries, uint32_t entries_cap,
                                                        uint64_t *__restrict__ vis) {
    __shared__ uint32_t base[kMaxBins], cnt[kMaxBins];
    __shared__ uint8_t fits[kMaxBins];
    const uint32_t tid = threadIdx.x;
    const int nbins = fp.nbx * fp.nby;
    const BinState &st = state[fp.set];
    const uint32_t rows_per_block = (fp.n_groups + 15u) / 16u;
    const uint32_t my_block = blockIdx.x / rows_per_block;
    const uint32_t n = surv_count[blockIdx.x];
    // where this workgroup's entries go inside each bin: start of the bin + entries of earlier row
    // blocks (boff) + entries of earlier workgroups of its own row block (scanned table)
    for (int b = int(tid); b < nbins; b += kSetupThreads) {
        const uint32_t s0 = st.start[b], t0 = st.total[b];
        base[b] = boff[size_t(my_block) * nbins + b] + dircount[size_t(blockIdx.x) * nbins + b];
        cnt[b] = 0;
        fits[b] = (s0 <= entries_cap && t0 <= entries_cap - s0) ? 1 : 0;
    }
    __syncthreads();
    const uint32_t loc_base = blockIdx.x * 2u * fp.chunk;
    for (uint32_t s = tid; s < n; s += kSetupThreads) {
        const uint32_t loc = loc_base + s;  // record slot; also what the visibility keys carry
        const uint2 sv = surv[loc];
        const int i0 = int(sv.x & 0xFFFFu), i1 = int(sv.x >> 16), j0 = int(sv.y & 0xFFFFu), j1 = int(sv.y >> 16);
        const int r0 = fp.rows - 1 - j1 - fp.ya, r1 = fp.rows - 1 - j0 - fp.ya;  // rows relative to the strip
        const int tx0 = i0 >> 6, tx1 = i1 >> 6, ty0 = r0 >> kTileShiftH, ty1 = r1 >> kTileShiftH;
        int bx0, bx1, by0, by1;
        bin_range(fp, i0, i1, j0, j1, bx0, bx1, by0, by1);
        bool direct = false;
        for (int by = by0; by <= by1; by++)
            for (int bx = bx0; bx <= bx1; bx++) {
                const int b = by * fp.nbx + bx;
                if (!fits[b]) { direct = true; continue; }
                // tiles of this bin inside the record's tile range -> bit mask, bit = ay * side + ax
                const int side = 1 << fp.bshift;
                const int ax0 = max(tx0 - side * bx, 0), ax1 = min(tx1 - side * bx, side - 1);
                const int ay0 = max(ty0 - side * by, 0), ay1 = min(ty1 - side * by, side - 1);
                const uint32_t rowbits = ((1u << (ax1 + 1)) - 1u) & ~((1u << ax0) - 1u);
                uint32_t m = 0;
                for (int ay = ay0; ay <= ay1; ay++) m |= rowbits << (side * ay);
                entries[size_t(base[b]) + atomicAdd(&cnt[b], 1u)] = loc | (m << kEntryMaskShift);
            }
        if (direct) {  // a bin it touches has no room: resolve through the global key buffer
            Edges e;
            edges_from_screen(records[loc], e);
            for (int j = j0; j <= j1; j++) {
                const size_t row = size_t(fp.rows - 1 - j) * fp.cols;
                for (int i = i0; i <= i1; i++) {
                    float f[3], zw;
                    if (cover(e, i, j, f, zw)) vis_min(vis, row + i, make_key(zw, loc));
                }
            }
        }
    }
}

// Colour of cell (i, j) for a record the near plane cut.  The record holds the sub-triangle's screen
// vertices, so only the varyings of its three vertices are missing: each is a mesh vertex or the point
// I + (O - I) * (dI / (dI - dO)) on the edge from an inside vertex I to an outside vertex O, exactly as
// clip_near builds them.  Which is which follows from the signs of d = z_c + w_c of the three mesh
// vertices; the varyings are then rebuilt and interpolated one component at a time, which keeps this
// rarely taken path from setting the register budget of the kernel that contains it.
template <bool TEX>
__device__ __forceinline__ void shade_clipped_record(const FrameParams &fp, const float *__restrict__ vbo,
                                                     const uint32_t *__restrict__ idx, const uint4 *__restrict__ triuv,
                                                     const uint32_t *__restrict__ atlas, const BinRec &rec, int i, int j,
                                                     uint8_t (&px)[4]) {
    const uint32_t t = rec.id >> 1, sub = rec.id & 1u;
    const float *vp[3];
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        vp[k] = vbo + size_t(idx[3 * size_t(t) + k]) * 9;
        const float x = vp[k][0], y = vp[k][1], z = vp[k][2];
        const float cz = ((fp.pv[2] * x + fp.pv[6] * y) + fp.pv[10] * z) + fp.pv[14];
        const float cw = ((fp.pv[3] * x + fp.pv[7] * y) + fp.pv[11] * z) + fp.pv[15];
        d[k] = cz + cw;
    }
    const bool in0 = d[0] >= 0.0f, in1 = d[1] >= 0.0f, in2 = d[2] >= 0.0f;
    const int n_in = int(in0) + int(in1) + int(in2);
    // A = the odd vertex out (the only inside one, or the only outside one), B and C follow it in winding order
    const int a = (n_in == 1) ? (in0 ? 0 : (in1 ? 1 : 2)) : (!in0 ? 0 : (!in1 ? 1 : 2));
    const int b = (a + 1) % 3, c = (a + 2) % 3;
    // sub-triangle vertex m = mesh vertex I[m], or the point on the edge I[m] -> O[m] when O[m] >= 0
    int I[3], O[3];
    if (n_in == 1) { I[0] = a; O[0] = -1; I[1] = a; O[1] = b; I[2] = a; O[2] = c; }
    else if (sub == 0u) { I[0] = b; O[0] = a; I[1] = b; O[1] = -1; I[2] = c; O[2] = -1; }
    else { I[0] = b; O[0] = a; I[1] = c; O[1] = -1; I[2] = c; O[2] = a; }
    float tt[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const float dI = I[m] == 0 ? d[0] : (I[m] == 1 ? d[1] : d[2]);
        const float dO = O[m] == 0 ? d[0] : (O[m] == 1 ? d[1] : d[2]);
        tt[m] = dI / (dI - dO);
    }
    Edges e;
    edges_from_screen(rec, e);
    const double pxd = double(i) + 0.5, pyd = double(j) + 0.5;
    float f[3];
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = float(e.a[k] * pxd + (e.b[k] * pyd + e.c[k]));
    const float rS = 1.0f / ((f[0] + f[1]) + f[2]);
    const float b0 = f[0] * rS, b1 = f[1] * rS, b2 = f[2] * rS;
    // one varying with values o0, o1, o2 at the mesh vertices, at this cell
    auto varying = [&](float o0, float o1, float o2) {
        float sv[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const float vi = I[m] == 0 ? o0 : (I[m] == 1 ? o1 : o2);
            const float vo = O[m] == 0 ? o0 : (O[m] == 1 ? o1 : o2);
            sv[m] = O[m] < 0 ? vi : vi + (vo - vi) * tt[m];
        }
        return (b0 * sv[0] + b1 * sv[1]) + b2 * sv[2];
    };
    float v[9];
    for (int q = 0; q < 9; q++) v[q] = varying(vp[0][q], vp[1][q], vp[2][q]);
    if (TEX && (rec.flags & kFlagTextured)) {
        const uint4 uv = triuv[t];
        const float tu = varying(float(uv.x & 0xFFFFu), float(uv.y & 0xFFFFu), float(uv.z & 0xFFFFu));
        const float tv = varying(float(uv.x >> 16), float(uv.y >> 16), float(uv.z >> 16));
        if (apply_texture(fp, atlas, tu, tv, v + 6, px)) return;
    }
    shade(fp, v, px);
}

// Colour and depth of cell (i, j) from the winner's setup record; unclipped records take normal and
// colour from the triangle's attribute row (one vertex's worth when the triangle is flat) and world
// positions from the position planes (only when point lights are on); clipped ones are rebuilt by
// shade_cell.
template <bool TEX>
__device__ __forceinline__ void shade_from_record(const FrameParams &fp, const float *__restrict__ vbo,
                                                  const uint32_t *__restrict__ idx, const float *__restrict__ triattr,
                                                  const float *__restrict__ tripos, size_t stride,
                                                  const uint4 *__restrict__ triuv, const uint32_t *__restrict__ atlas,
                                                  const BinRec &rec, int i, int j, uint8_t (&px)[4], float &zw) {
    if (rec.flags & kFlagClipped) { shade_clipped_record<TEX>(fp, vbo, idx, triuv, atlas, rec, i, j, px); return; }  // zw: the key's
    const uint32_t t = rec.id >> 1;
    const float4 *row = reinterpret_cast<const float4 *>(triattr) + size_t(t) * 6;
    const int o1 = (rec.flags & kFlagFlat) ? 0 : 2, o2 = (rec.flags & kFlagFlat) ? 0 : 4;
    const float4 a0 = row[0], a1 = row[o1], a2 = row[o2];
    const float2 c0 = *reinterpret_cast<const float2 *>(row + 1), c1 = *reinterpret_cast<const float2 *>(row + o1 + 1),
                 c2 = *reinterpret_cast<const float2 *>(row + o2 + 1);
    Edges e;
    edges_from_screen(rec, e);
    float f[3];
    cover(e, i, j, f, zw);
    const float rS = 1.0f / ((f[0] + f[1]) + f[2]);
    const float b0 = f[0] * rS, b1 = f[1] * rS, b2 = f[2] * rS;
    float v[9];
    v[3] = (b0 * a0.x + b1 * a1.x) + b2 * a2.x;
    v[4] = (b0 * a0.y + b1 * a1.y) + b2 * a2.y;
    v[5] = (b0 * a0.z + b1 * a1.z) + b2 * a2.z;
    v[6] = (b0 * a0.w + b1 * a1.w) + b2 * a2.w;
    v[7] = (b0 * c0.x + b1 * c1.x) + b2 * c2.x;
    v[8] = (b0 * c0.y + b1 * c1.y) + b2 * c2.y;
    if (fp.n_pl > 0) {
        float p[9];
#pragma unroll
        for (int q = 0; q < 9; q++) p[q] = tripos[size_t(q) * stride + t];
#pragma unroll
        for (int q = 0; q < 3; q++) v[q] = (b0 * p[q] + b1 * p[3 + q]) + b2 * p[6 + q];
    } else {
        v[0] = v[1] = v[2] = 0.0f;  // only the point-light loop reads the position
    }
    if (TEX && (rec.flags & kFlagTextured)) {
        const uint4 uv = triuv[t];
        const float tu = (b0 * float(uv.x & 0xFFFFu) + b1 * float(uv.y & 0xFFFFu)) + b2 * float(uv.z & 0xFFFFu);
        const float tv = (b0 * float(uv.x >> 16) + b1 * float(uv.y >> 16)) + b2 * float(uv.z >> 16);
        if (apply_texture(fp, atlas, tu, tv, v + 6, px)) return;
    }
    shade(fp, v, px);
}

// Scans rows [r0, r1] x columns [i0, i1] (at most 64 columns) of one record into the LDS key tile.
// Most cells of a cell range lie outside the triangle, and lanes of a wavefront walk ranges of very
// different sizes, so the scan is split in two: (1) per row, each edge function is evaluated in
// binary32 and solved for the column where it crosses a conservative error band, which gives the
// interval of columns that are not certainly outside - a run of bits in a 64-cell candidate mask;
// (2) only those cells pay for the binary64 evaluation that decides coverage and feeds the depth.
// The result is exactly that of testing every cell in binary64.
// Band: with a32 = fl(a), r32 = fl(b py + c), |a32 px + r32 - E| <= 2^-24 (|a| px + |r|) (1 + eps); a cell
// with E >= 0 therefore has a32 px + r32 >= -band / 32 for band = 2^-18 (|a32| px_max + |r32|), and lies
// beyond the crossing column x of "a32 px + r32 = -band" by at least 0.8 band / |a32| >= 2^-19 |x|, far more
// than the 2^-22 |x| the binary32 evaluation of x can be off.  Edges with a32 == 0 (or an overflowing
// reciprocal) put no bound on the interval.
__device__ __forceinline__ void raster_range(const FrameParams &fp, const Edges &ed, uint32_t key_low, int i0, int i1,
                                             int r0, int r1, int tx0, int tr0, uint64_t *keys) {
    constexpr float kBand = 1.0f / 262144.0f;  // 2^-18
    const float a32[3] = {float(ed.a[0]), float(ed.a[1]), float(ed.a[2])};
    // (hardware reciprocal: 1 ulp, inside the error budget above)
    const float inv_a[3] = {__builtin_amdgcn_rcpf(a32[0]), __builtin_amdgcn_rcpf(a32[1]), __builtin_amdgcn_rcpf(a32[2])};
    // coverage is E_k >= thr_k with thr_k = 0 on edges that own their boundary and the smallest positive
    // binary64 otherwise (E_k > 0): the same decision as edge_inside with one comparison per edge
    double thr[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        thr[k] = (ed.a[k] > 0.0 || (ed.a[k] == 0.0 && ed.b[k] > 0.0)) ? 0.0 : 4.9406564584124654e-324;
    const float apx[3] = {fabsf(a32[0]) * float(i1 + 1), fabsf(a32[1]) * float(i1 + 1), fabsf(a32[2]) * float(i1 + 1)};
    const int w = i1 - i0 + 1;
    const int rows_per_chunk = max(1, 64 / w);
    const float inv_w = 1.0f / float(w);
    const float flo = float(i0), fhi = float(i1);
    for (int rc = r0; rc <= r1; rc += rows_per_chunk) {
        const int rc1 = min(rc + rows_per_chunk - 1, r1);
        unsigned long long cand = 0ull;
        for (int rr = rc; rr <= rc1; rr++) {
            const double py = double(fp.rows - 1 - rr) + 0.5;
            float lo = flo, hi = fhi;  // candidate columns, as cell indices
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float r32 = float(ed.b[k] * py + ed.c[k]);
                const float band = kBand * (apx[k] + fabsf(r32));
                const float x = (-band - r32) * inv_a[k] - 0.5f;  // crossing column as a cell index
                lo = fmaxf(lo, a32[k] > 0.0f ? x : flo);
                hi = fminf(hi, a32[k] < 0.0f ? x : fhi);
            }
            if (!(lo <= hi)) continue;
            const int ilo = int(ceilf(lo)), ihi = int(floorf(hi));
            const int n = ihi - ilo + 1;
            if (n <= 0) continue;
            const unsigned long long run = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
            cand |= run << ((rr - rc) * w + (ilo - i0));
        }
        uint32_t clo = uint32_t(cand), chi = uint32_t(cand >> 32);
        while (clo | chi) {
            int k;
            if (clo) { k = __ffs(int(clo)) - 1; clo &= clo - 1u; }
            else { k = 31 + __ffs(int(chi)); chi &= chi - 1u; }
            const int dr = int((float(k) + 0.5f) * inv_w);  // k / w, exact for k < 64
            const int rr = rc + dr, i = i0 + (k - dr * w);
            const double px = double(i) + 0.5, py = double(fp.rows - 1 - rr) + 0.5;
            const double E0 = ed.a[0] * px + (ed.b[0] * py + ed.c[0]);
            const double E1 = ed.a[1] * px + (ed.b[1] * py + ed.c[1]);
            const double E2 = ed.a[2] * px + (ed.b[2] * py + ed.c[2]);
            const float f0 = float(E0), f1 = float(E1), f2 = float(E2);
            const float den = (f0 * ed.W[0] + f1 * ed.W[1]) + f2 * ed.W[2];
            const float num = (f0 * ed.Z[0] + f1 * ed.Z[1]) + f2 * ed.Z[2];
            const float zw = (num / den) * 0.5f + 0.5f;
            if ((E0 >= thr[0]) & (E1 >= thr[1]) & (E2 >= thr[2]) & (zw >= 0.0f) & (zw < 1.0f))
                atomicMin(reinterpret_cast<unsigned long long *>(&keys[(rr - tr0) * kTileW + (i - tx0)]),
                          (unsigned long long)make_key(zw, key_low));
        }
    }
}

// TEX: the frame samples an atlas (texel coordinates and atlas are only read then).
template <bool TEX>
__global__ __launch_bounds__(256) void k_tile(FrameParams fp, const float *__restrict__ vbo,
                                              const uint32_t *__restrict__ idx, const float *__restrict__ triattr,
                                              const float *__restrict__ tripos, size_t stride,
                                              const uint4 *__restrict__ triuv, const uint32_t *__restrict__ atlas,
                                              const BinRec *__restrict__ records,
                                              const uint32_t *__restrict__ entries, uint32_t entries_cap,
                                              const BinState *__restrict__ state, uint64_t *__restrict__ vis,
                                              uint8_t *__restrict__ rgba, float *__restrict__ depth,
                                              uint8_t *__restrict__ ascii, uint8_t *__restrict__ ridx,
                                              uint32_t *__restrict__ ids) {
    __shared__ uint64_t keys[kTileCells];
    __shared__ uint32_t list[kTileListMax];
    __shared__ uint32_t big[kTileBigMax];
    __shared__ uint16_t covered[kTileCells];
    __shared__ uint32_t n_list, n_big, n_cov;
    const uint32_t tid = threadIdx.x;
    // Workgroups are handed to the eight XCDs round-robin and each XCD has its own L2: the tiles of one
    // bin, which all read the bin's entry array, are given workgroup numbers that are equal modulo 8.
    // Workgroup B = 8 * tiles_per_bin * g + 8 * k + x is tile k of bin 8 g + x.
    const int side = 1 << fp.bshift, tiles_per_bin = side * side;
    const int wg_group = int(blockIdx.x) / (8 * tiles_per_bin), wg_in = int(blockIdx.x) % (8 * tiles_per_bin);
    const int bin = wg_group * 8 + (wg_in & 7), sub = wg_in >> 3;
    if (bin >= fp.nbx * fp.nby) return;
    const int tile_x = (bin % fp.nbx) * side + (sub & (side - 1)), tile_y = (bin / fp.nbx) * side + (sub >> fp.bshift);
    if (tile_x >= fp.ntx || tile_y >= fp.nty) return;
    const int tx0 = tile_x * kTileW, tr0 = fp.ya + tile_y * kTileH;                      // first column / output row
    const int tx1 = min(tx0 + kTileW, fp.cols) - 1, tr1 = min(tr0 + kTileH, fp.yb) - 1;  // last column / row
    const uint32_t bit = 1u << sub;
    const BinState &st = state[fp.set];
    for (int c = int(tid); c < kTileCells; c += 256) keys[c] = kBackgroundKey;
    if (tid == 0) { n_list = 0; n_big = 0; n_cov = 0; }
    const uint32_t e_begin = st.start[bin];
    const uint32_t e_total = st.total[bin];
    const bool stored = e_begin <= entries_cap && e_total <= entries_cap - e_begin;  // same test as k_fill
    const uint32_t e_end = stored ? e_begin + e_total : e_begin;
    __syncthreads();

    // filter the bin's entries by this tile's mask bit into the LDS list; whenever the list could
    // overflow on the next step (or at the end) rasterise what it holds
    // (kFilterBatch entries per lane and step, all loads in flight together: the step is one memory latency)
    for (uint32_t eb = e_begin;; eb += 256 * kFilterBatch) {
        uint32_t ent[kFilterBatch];
#pragma unroll
        for (int q = 0; q < kFilterBatch; q++) {
            const uint32_t e = eb + uint32_t(q) * 256u + tid;
            ent[q] = e < e_end ? entries[e] : 0u;  // mask 0 never matches
        }
#pragma unroll
        for (int q = 0; q < kFilterBatch; q++)
            if ((ent[q] >> kEntryMaskShift) & bit) list[atomicAdd(&n_list, 1u)] = ent[q] & kEntryLocMask;
        __syncthreads();
        const uint32_t nl = n_list;
        const bool last = eb + 256 * kFilterBatch >= e_end;
        if (nl + 256 * kFilterBatch > uint32_t(kTileListMax) || last) {
            // records with a small cell range inside the tile: one lane each; others -> sweep list
            for (uint32_t r = tid; r < nl; r += 256) {
                const uint32_t loc = list[r];
                const BinRec rec = records[loc];
                const int i0 = max(int(rec.i0), tx0), i1 = min(int(rec.i1), tx1);
                const int r0 = max(fp.rows - 1 - int(rec.j1), tr0), r1 = min(fp.rows - 1 - int(rec.j0), tr1);
                if ((i1 - i0 + 1) * (r1 - r0 + 1) > kTileSmallCells) {
                    const uint32_t slot = atomicAdd(&n_big, 1u);
                    if (slot < uint32_t(kTileBigMax)) { big[slot] = loc; continue; }
                    // list full: this lane scans the range itself (slow but correct)
                }
                Edges ed;
                edges_from_screen(rec, ed);
                raster_range(fp, ed, loc, i0, i1, r0, r1, tx0, tr0, keys);
            }
            __syncthreads();
            // sweep list: all lanes cover one record's cell range, 256 cells per step
            const uint32_t nb = min(n_big, uint32_t(kTileBigMax));
            for (uint32_t b = 0; b < nb; b++) {
                const uint32_t loc = big[b];
                const BinRec rec = records[loc];
                const int i0 = max(int(rec.i0), tx0), i1 = min(int(rec.i1), tx1);
                const int r0 = max(fp.rows - 1 - int(rec.j1), tr0), r1 = min(fp.rows - 1 - int(rec.j0), tr1);
                const int w = i1 - i0 + 1, n = w * (r1 - r0 + 1);
                Edges ed;
                edges_from_screen(rec, ed);
                for (int k = int(tid); k < n; k += 256) {
                    const int dr = k / w, i = i0 + (k - dr * w), rr = r0 + dr;
                    float f[3], zw;
                    if (cover(ed, i, fp.rows - 1 - rr, f, zw))
                        atomicMin(reinterpret_cast<unsigned long long *>(&keys[(rr - tr0) * kTileW + (i - tx0)]),
                                  (unsigned long long)make_key(zw, loc));
                }
            }
            __syncthreads();
            if (tid == 0) { n_list = 0; n_big = 0; }
            __syncthreads();
        }
        if (last) break;
    }

    // sub-triangles k_fill could not store were resolved through the global key buffer
    if (st.overflow != 0) {
        for (int c = int(tid); c < kTileCells; c += 256) {
            const int rr = tr0 + (c >> 6), i = tx0 + (c & 63);
            if (i > tx1 || rr > tr1) continue;
            const size_t cell = size_t(rr) * fp.cols + i;
            const uint64_t g = vis[cell];
            if (g != kBackgroundKey) {
                vis[cell] = kBackgroundKey;
                if (g < keys[c]) keys[c] = g;
            }
        }
        __syncthreads();
    }

    // compact the covered cells so that shading runs on full wavefronts
    for (int c = int(tid); c < kTileCells; c += 256)
        if (keys[c] != kBackgroundKey) covered[atomicAdd(&n_cov, 1u)] = uint16_t(c);
    __syncthreads();
    const uint32_t nc = n_cov;
    for (uint32_t k = tid; k < nc; k += 256) {
        const int c = int(covered[k]);
        const int rr = tr0 + (c >> 6), i = tx0 + (c & 63);
        const BinRec rec = records[uint32_t(keys[c])];  // the key's low word is the winner's record slot
        uint8_t px[4];
        float zw = __uint_as_float(uint32_t(keys[c] >> 32));  // the depth the raster pass computed from this record
        shade_from_record<TEX>(fp, vbo, idx, triattr, tripos, stride, triuv, atlas, rec, i, fp.rows - 1 - rr, px, zw);
        if (fp.want_ids) ids[size_t(rr) * fp.cols + i] = rec.id;
        keys[c] = (uint64_t(__float_as_uint(zw)) << 32) |
                  (uint32_t(px[0]) | (uint32_t(px[1]) << 8) | (uint32_t(px[2]) << 16) | (uint32_t(px[3]) << 24));
    }
    __syncthreads();

    // the tile's rows: 64 consecutive cells per row segment
    for (int c = int(tid); c < kTileCells; c += 256) {
        const int rr = tr0 + (c >> 6), i = tx0 + (c & 63);
        if (i > tx1 || rr > tr1) continue;
        const size_t cell = size_t(rr) * fp.cols + i;
        const uint64_t k = keys[c];
        const bool bg = k == kBackgroundKey;  // a shaded cell always has depth < 1.0 in the high word
        const uint32_t col = bg ? 0xFF000000u : uint32_t(k);
        reinterpret_cast<uint32_t *>(rgba)[cell] = col;
        depth[cell] = bg ? 1.0f : __uint_as_float(uint32_t(k >> 32));
        if (fp.want_ids && bg) ids[cell] = 0xFFFFFFFFu;
        const int ri = ramp_index(int(col & 255u), int((col >> 8) & 255u), int((col >> 16) & 255u), fp.ramp_len);
        if (fp.mode_on) ridx[cell] = uint8_t(ri);
        else if (rr >= fp.row0 && rr < fp.row1) {
            const uint32_t qa = col >> 24;  // 255 unless an atlas texel put an ASCII override there
            ascii[cell] = (TEX && qa >= 2u && qa <= 254u) ? uint8_t(qa) : fp.ramp[ri];
        }
    }
}

// ASCII stage without smoothing over an RGBA8 grid: alpha 2..254 is a literal code, otherwise the ramp.
__global__ __launch_bounds__(256) void k_quantize_plain(FrameParams fp, const uint32_t *__restrict__ px,
                                                        uint8_t *__restrict__ ascii, size_t n) {
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t me = px[p];
    const int a = int(me >> 24);
    ascii[p] = (a >= 2 && a <= 254)
                   ? uint8_t(a)
                   : fp.ramp[ramp_index(int(me & 255u), int((me >> 8) & 255u), int((me >> 16) & 255u), fp.ramp_len)];
}

// Text colour of the DOM path (js/renderer.js:74-84): channels snapped through a 256-entry table,
// packed (r << 16) | (g << 8) | b; alpha 2..254 (a UI cell, renderer.js:38-44) is black.
struct ColorLut {
    uint8_t q[256];
};

__global__ __launch_bounds__(256) void k_colors(ColorLut lut, const uint32_t *__restrict__ px, uint32_t *__restrict__ colors, size_t n) {
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t me = px[p];
    const uint32_t a = me >> 24;
    colors[p] = (a >= 2u && a <= 254u)
                    ? 0u
                    : (uint32_t(lut.q[me & 255u]) << 16) | (uint32_t(lut.q[(me >> 8) & 255u]) << 8) | uint32_t(lut.q[(me >> 16) & 255u]);
}

// Glyph compositing (js/ascii_pass_shader.js:190-230) onto a canvas of cols*cell_w x rows*cell_h pixels,
// row 0 = top.  A lane produces four consecutive pixels of one canvas row (one 16-byte store): per pixel the glyph's atlas
// texel (nearest; 2 pixels of tile padding skipped), coverage = cov[alpha] (the shader's pow(alpha,
// gamma), tabulated on the host), colour = mix(1, tint, coverage) per channel in binary32, bytes rounded
// like the raster stage's.  A streaming kernel: 4 bytes written per pixel, the per-cell inputs and the
// atlas are re-read from cache.
struct PresentParams {
    int cols, rows, cell_w, cell_h, W, H, atlas_w, tile_w, tile_h;
    int grayscale, transparent_bg, vec_store;
    float inv_cell_w;  // 1 / cell_w: (X + 0.5) * inv_cell_w truncates to X / cell_w for every X below 2^20
};

__global__ __launch_bounds__(256) void k_present(PresentParams pp, const uint32_t *__restrict__ px, const uint8_t *__restrict__ codes,
                                                 const uint8_t *__restrict__ atlas, const float *__restrict__ cov_of,
                                                 uint32_t *__restrict__ image) {
    __shared__ float cov[256], unit[256];  // coverage per alpha byte, colour byte / 255
    cov[threadIdx.x] = cov_of[threadIdx.x];
    unit[threadIdx.x] = float(threadIdx.x) / 255.0f;
    __syncthreads();
    const int X0 = (int(blockIdx.x) * 256 + int(threadIdx.x)) * 4, Y = int(blockIdx.y);
    if (X0 >= pp.W) return;
    const int cy = Y / pp.cell_h, py = Y - cy * pp.cell_h;  // uniform over the workgroup
    int cx = int((float(X0) + 0.5f) * pp.inv_cell_w), pxl = X0 - cx * pp.cell_w;
    uint32_t out[4];
    float tint[3] = {0.0f, 0.0f, 0.0f};
    const uint8_t *trow = nullptr;
    int loaded = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        out[k] = 0u;
        if (X0 + k < pp.W) {
            if (loaded != cx) {
                const size_t cell = size_t(cy) * pp.cols + cx;
                const uint32_t cell_px = px[cell];
                if (!pp.grayscale) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) tint[ch] = unit[(cell_px >> (8 * ch)) & 255u];
                }
                const int g = codes[cell];
                trow = atlas + size_t((g >> 4) * pp.tile_h + 2 + py) * pp.atlas_w + (g & 15) * pp.tile_w + 2;
                loaded = cx;
            }
            const float c = cov[trow[pxl]];
            if (!(pp.transparent_bg && c <= 0.0f)) {
                uint32_t v = 0xFF000000u;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float o = (1.0f - c) + tint[ch] * c;
                    o = fminf(fmaxf(o, 0.0f), 1.0f);
                    v |= uint32_t(floorf(o * 255.0f + 0.5f)) << (8 * ch);
                }
                out[k] = v;
            }
        }
        if (++pxl == pp.cell_w) { pxl = 0; cx++; }
    }
    uint32_t *dst = image + size_t(Y) * pp.W + X0;
    if (pp.vec_store) {
        *reinterpret_cast<uint4 *>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (X0 + k < pp.W) dst[k] = out[k];
    }
}

// Modal smoothing (ascii_pass_shader.js:77-138,169-185) over rows [row0, row1): a 32x8 tile of cells
// per workgroup, ramp indices of the tile and its halo of R cells staged in LDS (grid-clamped, as the
// shader's clampCell does), -1 marking alpha-override cells, which neither vote nor get smoothed.
// FROM_RGBA: src is an RGBA8 grid (AsciiPass.drawFromBuffer input); otherwise a plane of ramp
// indices written by k_shade (rendered frames carry no overrides).
template <int R, bool FROM_RGBA>
__global__ __launch_bounds__(256) void k_ascii(FrameParams fp, const void *__restrict__ src, uint8_t *__restrict__ ascii) {
    constexpr int TW = 32 + 2 * R, TH = 8 + 2 * R;
    __shared__ int16_t tile[TW * TH];
    const int x0 = int(blockIdx.x) * 32, y0 = fp.row0 + int(blockIdx.y) * 8;
    for (int k = int(threadIdx.x); k < TW * TH; k += 256) {
        const int ty = k / TW, tx = k - ty * TW;
        const int gx = min(max(x0 + tx - R, 0), fp.cols - 1), gy = min(max(y0 + ty - R, 0), fp.rows - 1);
        const size_t g = size_t(gy) * fp.cols + gx;
        int val;
        if (FROM_RGBA) {
            const uint32_t q = static_cast<const uint32_t *>(src)[g];
            const int qa = int(q >> 24);
            val = (qa >= 2 && qa <= 254) ? -1 : ramp_index(int(q & 255u), int((q >> 8) & 255u), int((q >> 16) & 255u), fp.ramp_len);
        } else {
            val = static_cast<const uint8_t *>(src)[g];
        }
        tile[k] = int16_t(val);
    }
    __syncthreads();
    const int lx = int(threadIdx.x & 31u), ly = int(threadIdx.x >> 5);
    const int x = x0 + lx, y = y0 + ly;
    if (x >= fp.cols || y >= fp.row1) return;
    const size_t cell = size_t(y) * fp.cols + x;
    const int base = tile[(ly + R) * TW + lx + R];
    if (FROM_RGBA && base < 0) {
        ascii[cell] = uint8_t(static_cast<const uint32_t *>(src)[cell] >> 24);
        return;
    }
    int nb[(2 * R + 1) * (2 * R + 1) - 1];
    {
        int q = 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++)
                if (dx != 0 || dy != 0) nb[q++] = tile[(ly + R + dy) * TW + lx + R + dx];
    }
    int cand = -1, cnt = 0, votes = 0;  // Boyer-Moore candidate in shader order, then its true vote count
#pragma unroll
    for (int q = 0; q < (2 * R + 1) * (2 * R + 1) - 1; q++) {
        const int ni = nb[q];
        if (ni < 0) continue;
        if (cnt == 0) { cand = ni; cnt = 1; }
        else if (ni == cand) cnt += 1;
        else cnt -= 1;
    }
#pragma unroll
    for (int q = 0; q < (2 * R + 1) * (2 * R + 1) - 1; q++) votes += (nb[q] >= 0 && nb[q] == cand) ? 1 : 0;
    const int fin = (cand >= 0 && votes >= fp.mode_thresh && cand != base) ? cand : base;
    ascii[cell] = (fin >= 0 && fin < fp.ramp_len) ? fp.ramp[fin] : uint8_t(32);
}

// Modal smoothing over a plane of ramp indices (what rendered frames produce: no override cells).
// A workgroup takes 128 x 8 cells, a lane four consecutive cells of one row.  The tile and its halo are
// staged as bytes with rows padded so that a lane's window (its four cells and R neighbours on either
// side) is three aligned dwords per row: 3 * (2R + 1) LDS reads serve four cells, and the neighbours are
// byte fields of registers.  Same Boyer-Moore order and vote rule as k_ascii.
template <int R>
__global__ __launch_bounds__(256) void k_ascii_plane(FrameParams fp, const uint8_t *__restrict__ src, uint8_t *__restrict__ ascii) {
    constexpr int TWD = 34, TH = 8 + 2 * R;  // dwords per tile row: 4 bytes of left halo, 128 cells, 4 of right halo
    __shared__ uint32_t tile[TWD * TH];
    __shared__ uint8_t ramp[AR_MAX_RAMP];
    const int x0 = int(blockIdx.x) * 128, y0 = fp.row0 + int(blockIdx.y) * 8;
    ramp[threadIdx.x] = fp.ramp[threadIdx.x];
    const bool cols4 = (fp.cols & 3) == 0;
    for (int k = int(threadIdx.x); k < TWD * TH; k += 256) {
        const int ty = k / TWD, td = k - ty * TWD;
        const int gy = min(max(y0 + ty - R, 0), fp.rows - 1);
        const int gx = x0 - 4 + 4 * td;  // column of the dword's first byte
        const uint8_t *row = src + size_t(gy) * fp.cols;
        uint32_t v;
        if (cols4 && gx >= 0 && gx + 3 < fp.cols) {
            v = *reinterpret_cast<const uint32_t *>(row + gx);
        } else {
            v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) v |= uint32_t(row[min(max(gx + q, 0), fp.cols - 1)]) << (8 * q);
        }
        tile[k] = v;
    }
    __syncthreads();
    const int lx = int(threadIdx.x & 31u), ly = int(threadIdx.x >> 5);
    const int x = x0 + 4 * lx, y = y0 + ly;
    if (x >= fp.cols || y >= fp.row1) return;
    uint32_t w[2 * R + 1][3];
#pragma unroll
    for (int dy = -R; dy <= R; dy++)
#pragma unroll
        for (int q = 0; q < 3; q++) w[dy + R][q] = tile[(ly + R + dy) * TWD + lx + q];
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // byte p of a window row is column x - 4 + p; cell j sits at p = 4 + j
        int cand = -1, cnt = 0, votes = 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                if (dx == 0 && dy == 0) continue;
                const int p = 4 + j + dx;
                const int ni = int((w[dy + R][p >> 2] >> (8 * (p & 3))) & 255u);
                cand = cnt == 0 ? ni : cand;
                cnt += ni == cand ? 1 : -1;
            }
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                if (dx == 0 && dy == 0) continue;
                const int p = 4 + j + dx;
                votes += int((w[dy + R][p >> 2] >> (8 * (p & 3))) & 255u) == cand ? 1 : 0;
            }
        const int base = int((w[R][(4 + j) >> 2] >> (8 * ((4 + j) & 3))) & 255u);
        const int fin = (votes >= fp.mode_thresh && cand != base) ? cand : base;
        out |= uint32_t(fin < fp.ramp_len ? ramp[fin] : uint8_t(32)) << (8 * j);
    }
    uint8_t *dst = ascii + size_t(y) * fp.cols + x;
    if (cols4) {
        *reinterpret_cast<uint32_t *>(dst) = out;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x + j < fp.cols) dst[j] = uint8_t(out >> (8 * j));
    }
}

template <bool FROM_RGBA>
void launch_ascii(const FrameParams &fp, const void *src, uint8_t *ascii, hipStream_t stream) {
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 3 ? 3 : fp.mode_radius);  // shader clamps to 1..3
    if constexpr (!FROM_RGBA) {
        const dim3 grid((fp.cols + 127) / 128, (fp.row1 - fp.row0 + 7) / 8);
        const uint8_t *plane = static_cast<const uint8_t *>(src);
        if (radius == 1) k_ascii_plane<1><<<grid, 256, 0, stream>>>(fp, plane, ascii);
        else if (radius == 2) k_ascii_plane<2><<<grid, 256, 0, stream>>>(fp, plane, ascii);
        else k_ascii_plane<3><<<grid, 256, 0, stream>>>(fp, plane, ascii);
    } else {
        const dim3 grid((fp.cols + 31) / 32, (fp.row1 - fp.row0 + 7) / 8);
        if (radius == 1) k_ascii<1, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
        else if (radius == 2) k_ascii<2, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
        else k_ascii<3, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
    }
}

// ------------------------------------------------------------------------------------------------
// Ray backend: the reference's deterministic single-bounce ray tracer (raytrace_shader.js:62-242) as one
// kernel, a lane per cell.  Binary32 throughout, no fused multiply-add, expressions left to right;
// normalize(v) = v * (1 / sqrt(dot(v, v))) with the zero vector staying zero, reflect(I, N) =
// I - (2 dot(N, I)) N, max / min = fmaxf / fminf.  The scene arrays are read at addresses that are the same
// for every lane (scalar loads); only the hit tests diverge.
// ------------------------------------------------------------------------------------------------
struct RtScene {  // device pointers into one allocation; counts as uploaded by raytrace.js:351-389
    int n_spheres, n_planes, n_tris, n_mats, n_pl, n_dl;
    const float *spheres4, *planes4, *tri_a, *tri_b, *tri_c, *mat_albedo, *pl_pos, *pl_col, *dl_dir, *dl_col;
    const int32_t *sphere_mat, *plane_mat, *tri_mat, *mat_reflective;
    float env[3];
};

struct RtCamera {  // per-frame terms of raytrace_shader.js:200-210, evaluated on the host
    float ro[3], uu[3], vv[3], ww[3];
    float focal, aspect;  // aspect = (cols / rows) * max(pixel aspect, 1e-6)
};

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 vload(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 vnorm(V3 a) {
    const float d = vdot(a, a);
    return d > 0.0f ? a * (1.0f / sqrtf(d)) : V3{0.0f, 0.0f, 0.0f};
}

constexpr float kRtEps = 1e-4f, kRtFar = 1e6f;

__device__ __forceinline__ float rt_sphere(V3 ro, V3 rd, const float *s) {  // raytrace_shader.js:89-101
    const V3 oc = ro - vload(s);
    const float b = vdot(oc, rd);
    const float c = vdot(oc, oc) - s[3] * s[3];
    const float h = b * b - c;
    if (h < 0.0f) return -1.0f;
    const float q = sqrtf(h);
    const float t1 = -b - q, t2 = -b + q;
    if (t1 > kRtEps) return t1;
    if (t2 > kRtEps) return t2;
    return -1.0f;
}

__device__ __forceinline__ float rt_plane(V3 ro, V3 rd, const float *pl) {  // :104-109
    const V3 n = vload(pl);
    const float denom = vdot(n, rd);
    if (fabsf(denom) < 1e-6f) return -1.0f;
    const float t = (-pl[3] - vdot(n, ro)) / denom;
    return t > kRtEps ? t : -1.0f;
}

__device__ __forceinline__ float rt_tri(V3 ro, V3 rd, V3 a, V3 b, V3 c, V3 &n) {  // :111-125
    const V3 e1 = b - a, e2 = c - a;
    const V3 p = vcross(rd, e2);
    const float det = vdot(e1, p);
    n = v3(0.0f, 0.0f, 0.0f);
    if (fabsf(det) < 1e-6f) return -1.0f;
    const float inv = 1.0f / det;
    const V3 t = ro - a;
    const float u = vdot(t, p) * inv;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const V3 q = vcross(t, e1);
    const float v = vdot(rd, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float tt = vdot(e2, q) * inv;
    if (tt <= kRtEps) return -1.0f;
    n = vnorm(vcross(e1, e2));
    if (vdot(n, rd) > 0.0f) n = v3(-n.x, -n.y, -n.z);
    return tt;
}

struct RtHit { float t; int m; V3 n; };

__device__ __forceinline__ RtHit rt_intersect(const RtScene &S, V3 ro, V3 rd) {  // :129-150
    RtHit H{kRtFar, -1, V3{0.0f, 0.0f, 0.0f}};
    for (int i = 0; i < S.n_spheres; i++) {
        const float *s = S.spheres4 + 4 * i;
        const float t = rt_sphere(ro, rd, s);
        if (t > kRtEps && t < H.t) {
            H.t = t; H.m = S.sphere_mat[i];
            const V3 pos = ro + rd * t;
            const float r = fmaxf(s[3], 1e-6f);
            const V3 d = pos - vload(s);
            H.n = v3(d.x / r, d.y / r, d.z / r);
        }
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float t = rt_plane(ro, rd, S.planes4 + 4 * i);
        if (t > kRtEps && t < H.t) { H.t = t; H.m = S.plane_mat[i]; H.n = vload(S.planes4 + 4 * i); }
    }
    for (int i = 0; i < S.n_tris; i++) {
        V3 n;
        const float t = rt_tri(ro, rd, vload(S.tri_a + 3 * i), vload(S.tri_b + 3 * i), vload(S.tri_c + 3 * i), n);
        if (t > kRtEps && t < H.t) { H.t = t; H.m = S.tri_mat[i]; H.n = n; }
    }
    return H;
}

__device__ __forceinline__ bool rt_occluded(const RtScene &S, V3 ro, V3 rd, float max_t) {  // :152-166: no planes
    for (int i = 0; i < S.n_spheres; i++) {
        const float t = rt_sphere(ro, rd, S.spheres4 + 4 * i);
        if (t > kRtEps && t < max_t) return true;
    }
    for (int i = 0; i < S.n_tris; i++) {
        V3 n;
        const float t = rt_tri(ro, rd, vload(S.tri_a + 3 * i), vload(S.tri_b + 3 * i), vload(S.tri_c + 3 * i), n);
        if (t > kRtEps && t < max_t) return true;
    }
    return false;
}

__device__ __forceinline__ V3 rt_shade_diffuse(const RtScene &S, V3 pos, V3 N, V3 albedo) {  // :168-196
    const float z = fmaxf(N.y * 0.0f, 0.0f);
    V3 Lo = v3(S.env[0] * z, S.env[1] * z, S.env[2] * z);
    const V3 org = pos + N * kRtEps;
    for (int i = 0; i < S.n_dl; i++) {
        const V3 d = vload(S.dl_dir + 3 * i);
        const V3 L = vnorm(v3(-d.x, -d.y, -d.z));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, 1e5f)) {
            const float *c = S.dl_col + 3 * i;
            Lo = v3(Lo.x + (albedo.x * c[0]) * ndl, Lo.y + (albedo.y * c[1]) * ndl, Lo.z + (albedo.z * c[2]) * ndl);
        }
    }
    for (int i = 0; i < S.n_pl; i++) {
        const V3 Lv = vload(S.pl_pos + 3 * i) - pos;
        const float dist2 = fmaxf(vdot(Lv, Lv), 1e-6f);
        const V3 L = Lv * (1.0f / sqrtf(dist2));
        const float ndl = fmaxf(vdot(N, L), 0.0f);
        if (ndl > 0.0f && !rt_occluded(S, org, L, sqrtf(dist2) - 2.0f * kRtEps)) {
            const float att = 1.0f / (1.0f + dist2 * 0.05f);
            const float w = ndl * att;
            const float *c = S.pl_col + 3 * i;
            Lo = v3(Lo.x + (albedo.x * c[0]) * w, Lo.y + (albedo.y * c[1]) * w, Lo.z + (albedo.z * c[2]) * w);
        }
    }
    return Lo;
}

__device__ __forceinline__ void rt_material(const RtScene &S, int m, V3 &albedo, bool &reflective) {  // :71-86,218-220
    const int mi = int(fminf(fmaxf(float(m), 0.0f), float(S.n_mats - 1)));
    albedo = v3(0.8f, 0.8f, 0.8f);
    reflective = false;
    if (mi >= 0 && mi < S.n_mats) { albedo = vload(S.mat_albedo + 3 * mi); reflective = S.mat_reflective[mi] != 0; }
}

// One lane per cell of output rows [ya, yb): primary ray, nearest hit, direct light with hard shadows, or one
// mirror bounce (raytrace_shader.js:198-242).  RGBA8, alpha 255, row 0 = top.
__global__ __launch_bounds__(256) void k_raytrace(RtScene S, RtCamera cam, int cols, int rows, int ya, int yb,
                                                  uint32_t *__restrict__ rgba) {
    const size_t n = size_t(yb - ya) * cols;
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const int r = ya + int(p / size_t(cols)), i = int(p % size_t(cols));
    const int j = rows - 1 - r;  // GL window row
    float px = -1.0f + 2.0f * ((float(i) + 0.5f) / float(cols));
    const float py = -1.0f + 2.0f * ((float(j) + 0.5f) / float(rows));
    px = px * cam.aspect;
    const V3 ro = vload(cam.ro), uu = vload(cam.uu), vv = vload(cam.vv), ww = vload(cam.ww);
    const V3 rd = vnorm(v3((px * uu.x + py * vv.x) + cam.focal * ww.x, (px * uu.y + py * vv.y) + cam.focal * ww.y,
                           (px * uu.z + py * vv.z) + cam.focal * ww.z));
    V3 col;
    const RtHit H = rt_intersect(S, ro, rd);
    if (H.m < 0) {
        col = vload(S.env);
    } else {
        const V3 hit = ro + rd * H.t;
        const V3 N = vnorm(H.n);
        V3 albedo; bool refl;
        rt_material(S, H.m, albedo, refl);
        if (!refl) {
            col = rt_shade_diffuse(S, hit, N, albedo);
        } else {
            const float d2 = 2.0f * vdot(N, rd);
            const V3 rdir = rd - N * d2;
            const V3 org = hit + N * kRtEps;
            const RtHit H2 = rt_intersect(S, org, rdir);
            if (H2.m < 0) {
                col = vload(S.env);
            } else {
                const V3 hit2 = org + rdir * H2.t;
                const V3 N2 = vnorm(H2.n);
                V3 alb2; bool r2;
                rt_material(S, H2.m, alb2, r2);
                col = rt_shade_diffuse(S, hit2, N2, alb2);
            }
        }
    }
    const float c3[3] = {col.x, col.y, col.z};
    uint32_t out = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) out |= uint32_t(int(floorf(fminf(fmaxf(c3[k], 0.0f), 1.0f) * 255.0f + 0.5f))) << (8 * k);
    rgba[size_t(r) * cols + i] = out;
}

thread_local char g_create_error[256] = "";

}  // namespace

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
    int overlap = 2;                // 0: frames one after the other on `stream`; 1, 2: consecutive frames alternate streams
    hipEvent_t ev_out[kSets + 1] = {};  // outputs complete on each frame stream / on `stream` (ar_stream_wait_frame)
    hipEvent_t ev_ext = nullptr;    // a caller's stream has consumed the outputs (ar_frame_wait_stream)
    float *d_vbo = nullptr;
    uint32_t *d_idx = nullptr;
    uint32_t n_verts = 0, n_tris = 0;
    float proj[16], view[16];
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
    float *d_tripos = nullptr;       // 9 planes x tripos_stride floats: triangle positions
    size_t tripos_stride = 0;
    float *d_triattr = nullptr;      // n_tris x 24 floats: the three vertices' normals and colours
    // everything the binning kernels write, per buffer set
    BinRec *d_records[kSets] = {};  // 2 x n_tris setup record slots, filled densely per k_setup workgroup
    uint8_t *d_flat = nullptr;       // n_tris flag bytes: kFlagFlat | kFlagTextured
    void *d_rt = nullptr;            // ray backend: the scene arrays of ar_rt_set_scene in one allocation
    RtScene rt_scene = {};           // counts and device pointers into d_rt
    bool rt_has_scene = false;
    uint4 *d_triuv = nullptr;        // n_tris texel-coordinate rows (ar_set_mesh_uv), else null
    uint32_t *d_atlas = nullptr;     // atlas texels (ar_set_texture), else null
    int atlas_w = 0, atlas_h = 0;
    uint2 *d_surv[kSets] = {};  // per k_setup workgroup: cell ranges of its surviving sub-triangles
    uint32_t *d_surv_count[kSets] = {};
    uint32_t *d_dircount[kSets] = {};  // per k_setup workgroup: entries per bin (+ scan scratch)
    size_t group_cap = 0, dircount_cap = 0;
    uint32_t *d_entries[kSets] = {};  // (record slot, tile mask) entries grouped by bin
    uint32_t pool_cap = 0;
    BinState *d_state = nullptr;     // kSets blocks
    uint32_t frame = 0;              // tiled frames enqueued so far
    bool last_tiled = false;
    uint8_t *d_q_rgba = nullptr, *d_q_ascii = nullptr;
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
    (void)hipFree(ctx->d_flat); (void)hipFree(ctx->d_triuv);
    for (int k = 0; k < kSets; k++) { (void)hipFree(ctx->d_records[k]); ctx->d_records[k] = nullptr; }
    ctx->d_vbo = nullptr; ctx->d_idx = nullptr; ctx->d_tripos = nullptr; ctx->d_triattr = nullptr;
    ctx->d_flat = nullptr; ctx->d_triuv = nullptr;  // texel coordinates belong to the mesh they were given for
    ctx->n_verts = 0; ctx->n_tris = 0;
    return 0;
}

// d_vbo / d_idx hold the new mesh: derive the streams of the tiled path (positions, attribute rows, flag
// bytes) and allocate its record slots.
int finish_mesh(ar_ctx *ctx, uint32_t n_verts, uint32_t n_tris) {
    ctx->tripos_stride = (size_t(n_tris) + 63) / 64 * 64;
    AR_HIP(ctx, hipMalloc(&ctx->d_tripos, 9 * ctx->tripos_stride * sizeof(float)));
    AR_HIP(ctx, hipMalloc(&ctx->d_triattr, size_t(n_tris) * 24 * sizeof(float)));
    for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipMalloc(&ctx->d_records[k], 2 * size_t(n_tris) * sizeof(BinRec)));
    AR_HIP(ctx, hipMalloc(&ctx->d_flat, size_t(n_tris)));
    k_tristream<<<(n_tris + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_vbo, ctx->d_idx, n_verts, n_tris, ctx->tripos_stride, ctx->d_tripos,
                                                              ctx->d_triattr, ctx->d_flat);
    AR_HIP(ctx, hipGetLastError());
    AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_verts = n_verts; ctx->n_tris = n_tris;
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
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_targets(ctx, cols, rows);
    if (rc) return rc;
    if (ctx->n_tris == 0 || !ctx->d_vbo || !ctx->d_idx) return 0;  // raster.js:274: nothing to draw

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
    // texturing needs texel coordinates, an atlas, and more than one texel either way (shader_utils.js:94)
    const bool textured = ctx->d_triuv && ctx->d_atlas && ctx->atlas_w > 1 && ctx->atlas_h > 1;
    fp.tex_w = textured ? ctx->atlas_w : 0;
    fp.tex_h = textured ? ctx->atlas_h : 0;

    const size_t strip_cells = size_t(fp.yb - fp.ya) * cols;
    const unsigned cell_blocks = unsigned((strip_cells + 255) / 256);
    fp.ntx = (cols + kTileW - 1) / kTileW;
    fp.nty = (fp.yb - fp.ya + kTileH - 1) / kTileH;
    fp.bshift = 1;  // per-tile bins (0) shorten k_tile's filter but cost as much again in the binning kernels
    fp.nbx = (fp.ntx + (1 << fp.bshift) - 1) >> fp.bshift;
    fp.nby = (fp.nty + (1 << fp.bshift) - 1) >> fp.bshift;
    const size_t nt = size_t(fp.ntx) * fp.nty;
    const size_t nbins = size_t(fp.nbx) * fp.nby;
    const bool tiled = ctx->path == 0 && nt <= size_t(kMaxTiles) && nbins <= size_t(kMaxBins) && cols <= 65535 &&
                       rows <= 65535 && ctx->d_tripos && ctx->n_tris <= (kEntryLocMask >> 1) - 65536u;  // slots fit an entry
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
        // one triangle per k_setup lane and pass; more passes only beyond kMaxGroups workgroups
        uint32_t chunk = (fp.n_tris + kMaxGroups - 1) / kMaxGroups;
        chunk = (chunk + kSetupThreads - 1) / kSetupThreads * kSetupThreads;
        fp.chunk = chunk;
        fp.n_groups = (fp.n_tris + chunk - 1) / chunk;
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
        const size_t want_groups = size_t(fp.n_groups) * 2 * chunk;
        if (!ctx->d_surv[0] || ctx->group_cap < want_groups) {
            rc = sync_streams(ctx);
            if (rc) return rc;
            for (int k = 0; k < kSets; k++) {
                (void)hipFree(ctx->d_surv[k]); (void)hipFree(ctx->d_surv_count[k]);
                ctx->d_surv[k] = nullptr; ctx->d_surv_count[k] = nullptr;
            }
            ctx->group_cap = 0;
            for (int k = 0; k < kSets; k++) {
                AR_HIP(ctx, hipMalloc(&ctx->d_surv[k], want_groups * sizeof(uint2)));
                AR_HIP(ctx, hipMalloc(&ctx->d_surv_count[k], (size_t(kMaxGroups) + 1) * sizeof(uint32_t)));
            }
            ctx->group_cap = want_groups;
        }
        const size_t want_dir = size_t(fp.n_groups) * nbins;
        if (!ctx->d_dircount[0] || ctx->dircount_cap < want_dir) {
            rc = sync_streams(ctx);
            if (rc) return rc;
            for (int k = 0; k < kSets; k++) { (void)hipFree(ctx->d_dircount[k]); ctx->d_dircount[k] = nullptr; }
            ctx->dircount_cap = 0;
            for (int k = 0; k < kSets; k++) AR_HIP(ctx, hipMalloc(&ctx->d_dircount[k], (want_dir + 32 * size_t(kMaxBins)) * sizeof(uint32_t)));
            ctx->dircount_cap = want_dir;
        }
    }
    const bool was_clean = ctx->vis_clean;
    ctx->vis_clean = false;  // stays false if anything below fails: the next frame then starts from scratch
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
        uint32_t *blocksum = dircount + ctx->dircount_cap, *boff = blocksum + 16 * size_t(kMaxBins);
        {
            Stamper s(ctx, K_SETUP, fs);
            k_setup<<<fp.n_groups, kSetupThreads, 0, fs>>>(fp, ctx->d_tripos, ctx->tripos_stride, ctx->d_flat, ctx->d_records[set],
                                                         ctx->d_surv[set], ctx->d_surv_count[set], dircount);
        }
        {
            Stamper s(ctx, K_BINSCAN, fs);
            k_binscan<<<dim3(unsigned((nbins + 63) / 64), 16), 1024, 0, fs>>>(fp, dircount, blocksum);
        }
        {
            Stamper s(ctx, K_BINOFF, fs);
            k_binoffsets<<<1, 1024, 0, fs>>>(fp, blocksum, boff, ctx->d_state, ctx->pool_cap);
        }
        {
            Stamper s(ctx, K_FILL, fs);
            k_fill<<<fp.n_groups, kSetupThreads, 0, fs>>>(fp, ctx->d_records[set], ctx->d_surv[set], ctx->d_surv_count[set], dircount, boff,
                                                        ctx->d_state, ctx->d_entries[set], ctx->pool_cap, ctx->d_vis[set]);
        }
        {
            Stamper s(ctx, K_TILE, fs);
            const unsigned grid = unsigned((nbins + 7) / 8 * 8) << (2 * fp.bshift);
            auto kernel = textured ? k_tile<true> : k_tile<false>;
            kernel<<<grid, 256, 0, fs>>>(fp, ctx->d_vbo, ctx->d_idx, ctx->d_triattr, ctx->d_tripos, ctx->tripos_stride, ctx->d_triuv,
                                         ctx->d_atlas, ctx->d_records[set], ctx->d_entries[set], ctx->pool_cap, ctx->d_state,
                                         ctx->d_vis[set], ctx->d_rgba[oset], ctx->d_depth[oset], ascii_out, ctx->d_ridx[set],
                                         ctx->d_ids[oset]);
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
        if (textured) launch_ascii<true>(fp, ctx->d_rgba[oset], ascii_out, fs);
        else launch_ascii<false>(fp, ctx->d_ridx[tiled ? set : 0], ascii_out, fs);
    }
    AR_HIP(ctx, hipGetLastError());
    ctx->vis_clean = true;
    ctx->rendered = true;
    ctx->last_tiled = tiled;
    ctx->last_oset = oset;
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
    if (!ok) {
        fail(nullptr, AR_E_HIP, "context setup failed: %s", hipGetErrorString(hipGetLastError()));
        ar_destroy(ctx);
        return nullptr;
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
    (void)hipFree(ctx->d_vbo); (void)hipFree(ctx->d_idx);
    (void)hipFree(ctx->d_medium); (void)hipFree(ctx->d_large); (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii); (void)hipFree(ctx->d_q_colors);
    (void)hipFree(ctx->d_p_atlas); (void)hipFree(ctx->d_p_cov); (void)hipFree(ctx->d_p_image);
    (void)hipFree(ctx->d_state); (void)hipFree(ctx->d_tripos); (void)hipFree(ctx->d_triattr); (void)hipFree(ctx->d_flat);
    (void)hipFree(ctx->d_triuv); (void)hipFree(ctx->d_atlas); (void)hipFree(ctx->d_rt);
    for (int k = 0; k < kSets; k++) {
        (void)hipFree(ctx->d_entries[k]); (void)hipFree(ctx->d_surv[k]); (void)hipFree(ctx->d_surv_count[k]);
        (void)hipFree(ctx->d_dircount[k]); (void)hipFree(ctx->d_records[k]);
    }
    for (int k = 0; k <= kSets; k++)
        if (ctx->ev_out[k]) (void)hipEventDestroy(ctx->ev_out[k]);
    if (ctx->ev_ext) (void)hipEventDestroy(ctx->ev_ext);
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
    k_uvflags<<<(ctx->n_tris + 255) / 256, 256, 0, ctx->stream>>>(ctx->d_triuv, ctx->n_tris, ctx->d_flat);
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

int ar_sync(ar_ctx *ctx) {
    if (!ctx) return AR_E_ARG;
    AR_HIP(ctx, hipSetDevice(ctx->device));
    return sync_streams(ctx);
}

int ar_read(ar_ctx *ctx, uint8_t *rgba, float *depth, uint8_t *ascii, uint32_t *ids) {
    if (!ctx) return AR_E_ARG;
    if (!ctx->rendered) return fail(ctx, AR_E_STATE, "ar_read: no frame has been rendered at this grid size");
    AR_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_streams(ctx);
    if (rc) return rc;
    const size_t n = size_t(ctx->cols) * ctx->rows;
    const int o = ctx->last_oset;
    if (rgba) AR_HIP(ctx, hipMemcpy(rgba, ctx->d_rgba[o], n * 4, hipMemcpyDeviceToHost));
    if (depth) AR_HIP(ctx, hipMemcpy(depth, ctx->d_depth[o], n * sizeof(float), hipMemcpyDeviceToHost));
    if (ascii) AR_HIP(ctx, hipMemcpy(ascii, ctx->d_ascii[o], n, hipMemcpyDeviceToHost));
    if (ids) {
        if (!ctx->d_ids[o]) return fail(ctx, AR_E_STATE, "ar_read: ids wanted but ar_set_id_output was not on for the last frame");
        AR_HIP(ctx, hipMemcpy(ids, ctx->d_ids[o], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int ar_render(ar_ctx *ctx, int cols, int rows, uint8_t *rgba, float *depth, uint8_t *ascii) {
    if (!ctx) return AR_E_ARG;
    int rc = enqueue_frame(ctx, cols, rows, false);
    if (rc) return rc;
    if (ctx->n_tris == 0 || !ctx->rendered) return ar_sync(ctx);  // no geometry: outputs stay untouched (raster.js:274)
    return ar_read(ctx, rgba, depth, ascii, nullptr);
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
        (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii);
        ctx->d_q_rgba = nullptr; ctx->d_q_ascii = nullptr; ctx->q_cells = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_q_rgba, n * 4));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_ascii, n));
        ctx->q_cells = n;
    }
    FrameParams fp = ctx->fp;
    fp.cols = cols; fp.rows = rows; fp.row0 = 0; fp.row1 = rows; fp.ya = 0; fp.yb = rows;
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_q_rgba, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
    {
        Stamper s(ctx, K_ASCII);
        if (fp.mode_on == 1) launch_ascii<true>(fp, ctx->d_q_rgba, ctx->d_q_ascii, ctx->stream);
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
            launch_ascii<true>(fp, ctx->d_rgba[0], ascii_out, ctx->stream);
        } else {
            const size_t first = size_t(row0) * cols, n = size_t(row1 - row0) * cols;
            k_quantize_plain<<<unsigned((n + 255) / 256), 256, 0, ctx->stream>>>(
                fp, reinterpret_cast<const uint32_t *>(ctx->d_rgba[0]) + first, ascii_out + first, n);
        }
    }
    AR_HIP(ctx, hipGetLastError());
    ctx->rendered = true;
    ctx->last_oset = 0;
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

// d_rgba (device, n cells) -> colors (host).  steps <= 0: the reference's max(3, max(2, ramp length)).
int colors_from_device(ar_ctx *ctx, const uint8_t *d_rgba, size_t n, int steps, int grayscale, uint32_t *colors) {
    if (steps <= 0) {
        const int rl = ctx->fp.ramp_len < 2 ? 2 : ctx->fp.ramp_len;
        steps = rl < 3 ? 3 : rl;   // js/renderer.js:13-19
    }
    if (steps < 2 || steps > 256) return fail(ctx, AR_E_ARG, "colour steps %d not in 2..256", steps);
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
    if (n > ctx->q_cells) {
        AR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_q_rgba); (void)hipFree(ctx->d_q_ascii);
        ctx->d_q_rgba = nullptr; ctx->d_q_ascii = nullptr; ctx->q_cells = 0;
        AR_HIP(ctx, hipMalloc(&ctx->d_q_rgba, n * 4));
        AR_HIP(ctx, hipMalloc(&ctx->d_q_ascii, n));
        ctx->q_cells = n;
    }
    AR_HIP(ctx, hipMemcpyAsync(ctx->d_q_rgba, rgba, n * 4, hipMemcpyHostToDevice, ctx->stream));
    return colors_from_device(ctx, ctx->d_q_rgba, n, steps, grayscale, colors);
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
            if (value < 0 || value > 2) return fail(ctx, AR_E_ARG, "AR_OPT_OVERLAP takes 0, 1 or 2");
            int rc = sync_streams(ctx);
            if (rc) return rc;
            ctx->overlap = int(value);
            return 0;
        }
        case AR_OPT_POOL_RECORDS:
            if (value < 0 || value > 0x7FFFFFFFll) return fail(ctx, AR_E_ARG, "AR_OPT_POOL_RECORDS out of range");
            ctx->pool_cap_option = uint32_t(value);
            return 0;
        default:
            return fail(ctx, AR_E_ARG, "unknown option %d", option);
    }
}

}  // extern "C"
