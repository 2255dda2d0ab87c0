// raster_common.h - shared by both raster paths: frame parameters, vertex transform, near clip, edge functions, coverage test, visibility keys
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

constexpr uint64_t kBackgroundKey = (uint64_t(0x3F800000u) << 32) | 0xFFFFFFFFull;  // depth 1.0f
constexpr int kSmallMaxCells = 64;      // cell-range area a single lane scans itself
constexpr int kMediumMaxCells = 4096;   // above this a record goes to the per-cell pass
constexpr uint32_t kMediumCapacity = 1u << 18;
constexpr uint32_t kLargeCapacity = 1u << 14;

enum KernelSlot { K_CLEAR = 0, K_RASTER, K_MEDIUM, K_LARGE, K_SHADE, K_ASCII, K_SETUP, K_BINSCAN, K_FILL, K_TILE, K_PRESENT, K_RAYTRACE };
const char *const kKernelNames[AR_N_KERNELS] = {"k_clear",  "k_raster", "k_raster_medium", "k_raster_large",
                                                "k_shade", "k_ascii",  "k_setup",         "k_binscan",
                                                "k_fill", "k_tile", "k_present",
                                                "k_raytrace"};

// Everything a frame's kernels need besides the big buffers; passed by value as a kernel argument.
struct FrameParams {
    float pv[16];  // P*V (times the model matrix when one is set), column-major
    float model[12];  // model matrix without its bottom row: model[3 * c + r] = M[c][r], r < 3 (ar_set_model)
    int has_model;    // 0: no model matrix (world position and normal are the vertex's own)
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
    int tshift;      // log2 of the tile height in rows (5, 4 or 3)
    int id_bits;     // bits of a bin entry that hold 2*slot + sub (enough for the mesh) ...
    int cls_bits;    // ... and, above them, so many bits of work class (0..6; what is left below the tile mask)
    int ascii_path;  // smoothing kernel: 0 the fastest that covers the parameters, 1 never the bit-sliced one, 2 only the generic one,
                     // 3 the bit-sliced one wherever it covers the parameters
    int fold_scan;   // 1: no k_binscan in this frame - k_setup adds up the totals per bin and row block, k_fill sums the
                     // table rows before its own (frames with few bins, where that is cheaper than a kernel)
    int bin_starts;  // 1: k_binstart has run - the bins' starts and the row-block prefixes are in place, k_fill only adds them up
    uint32_t chunk, n_groups;  // triangles per k_setup workgroup, number of those workgroups
    uint32_t frame;  // frame counter
    int set;         // buffer set of this frame (frame % kSets): binning state block, survivor slots, ...
    int tex_w, tex_h;  // atlas size in texels; 0 x 0 while texturing is off
    uint8_t ramp[AR_MAX_RAMP];
};

// Several frames in one launch chain (ar_render_batch_async): the kernels of the tiled path take the frame from blockIdx.z,
// its parameters from an array in device memory and every per-frame buffer at a frame's stride from the first frame's.
struct BatchStrides {
    size_t surv;      // survivor places (cell-range tuples, ids) per frame
    size_t groups;    // survivor counters per frame
    size_t dircount;  // words of count table + row-block totals per frame
    size_t entries;   // bin entries per frame
    size_t cells;     // cells per frame: visibility keys, ramp-index plane, RGBA (x 4 bytes), depth, ids
    size_t ascii;     // bytes from one frame's ASCII plane to the next (the context's planes, or the caller's target)
    uint32_t frames;  // frames in the launch
};

// Setup record of one (sub-)triangle: edge functions, per-vertex w and clip z, cell range, key id.
struct TriRec {
    double a[3], b[3], c[3];
    float W[3], Z[3];
    int i0, i1, j0, j1;  // inclusive cell range, GL window rows (y up)
    uint32_t id;         // 2*triangle + sub-triangle
    uint32_t pad;
};

// The frame parameters where the launch put them: the kernel-argument segment, of which FrameParams is the first
// member in every kernel that calls this.  The compiler otherwise loads every field a kernel uses once, at its top,
// and keeps it in a scalar register to the end - more than there are, so they are parked in lanes of a vector
// register and fetched back with a vector instruction at each use.  Reading a late phase's constants through this
// pointer - opaque to the optimiser, so the loads stay where they are written - costs scalar loads instead.
using KernargFrameParams = const __attribute__((address_space(4))) FrameParams;
__device__ __forceinline__ KernargFrameParams *kernarg_params() {
    KernargFrameParams *q = (KernargFrameParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

// World position M * (x, y, z, 1) and world normal mat3(M) * n of a vertex under the model matrix: what a vertex
// shader with a uModel uniform hands on as v_pos and v_nrm.
template <typename FP>
__device__ __forceinline__ void model_point(const FP &fp, float x, float y, float z, float *o) {
#pragma unroll
    for (int r = 0; r < 3; r++) o[r] = ((fp.model[r] * x + fp.model[3 + r] * y) + fp.model[6 + r] * z) + fp.model[9 + r];
}
template <typename FP>
__device__ __forceinline__ void model_normal(const FP &fp, float x, float y, float z, float *o) {
#pragma unroll
    for (int r = 0; r < 3; r++) o[r] = (fp.model[r] * x + fp.model[3 + r] * y) + fp.model[6 + r] * z;
}

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
        if (fp.has_model) {
            model_point(fp, x, y, z, o.v);
            model_normal(fp, p[3], p[4], p[5], o.v + 3);
        }
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
    // all twelve coordinates finite: x * 0 is a zero for finite x and a NaN otherwise, so the sum of the twelve products
    // is zero exactly when all are finite (twelve plain multiply-adds and one comparison instead of twelve class tests)
    float z = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) z = __builtin_fmaf(c2[q], 0.0f, __builtin_fmaf(c1[q], 0.0f, __builtin_fmaf(c0[q], 0.0f, z)));
    const bool ok = (z == 0.0f) && (c0[3] > 0.0f) && (c1[3] > 0.0f) && (c2[3] > 0.0f);
    if (!ok) return false;
    s.W[0] = c0[3]; s.W[1] = c1[3]; s.W[2] = c2[3];
    s.Z[0] = c0[2]; s.Z[1] = c1[2]; s.Z[2] = c2[2];
    s.X[0] = (c0[0] + c0[3]) * hw; s.X[1] = (c1[0] + c1[3]) * hw; s.X[2] = (c2[0] + c2[3]) * hw;
    s.Y[0] = (c0[1] + c0[3]) * hh; s.Y[1] = (c1[1] + c1[3]) * hh; s.Y[2] = (c2[1] + c2[3]) * hh;
    return true;
}

// p q - r s in binary64 for binary32 inputs.  Both products are exact in binary64, so the fused form rounds the
// same exact difference once, as "round(pq) - round(rs)" does: one instruction less, the same bits.
__device__ __forceinline__ double diff_of_products(float p, float q, float r, float s) {
    return __builtin_fma(double(p), double(q), -(double(r) * double(s)));
}

// Edge functions e_k = v_{k+1} x v_{k+2} in binary64 (every product of two binary32 values is exact).
template <typename S>
__device__ __forceinline__ void edges_from_screen(const S &s, Edges &e) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        e.a[k] = diff_of_products(s.Y[p], s.W[q], s.Y[q], s.W[p]);
        e.b[k] = diff_of_products(s.X[q], s.W[p], s.X[p], s.W[q]);
        e.c[k] = diff_of_products(s.X[p], s.Y[q], s.X[q], s.Y[p]);
        e.W[k] = s.W[k];
        e.Z[k] = s.Z[k];
    }
}

// det = v0 . (v1 x v2) > 0: counter-clockwise in the window, i.e. front facing; <= 0 is culled
// (back face or zero area).
__device__ __forceinline__ double facing_det(const Screen &s) {
    const double a0 = diff_of_products(s.Y[1], s.W[2], s.Y[2], s.W[1]);
    const double b0 = diff_of_products(s.X[2], s.W[1], s.X[1], s.W[2]);
    const double c0 = diff_of_products(s.X[1], s.Y[2], s.X[2], s.Y[1]);
    return (double(s.X[0]) * a0 + double(s.Y[0]) * b0) + double(s.W[0]) * c0;
}
__device__ __forceinline__ bool front_facing(const Screen &s) { return facing_det(s) > 0.0; }

// Inclusive cell range (GL window coordinates) that contains every covered cell centre, clamped to
// the viewport columns and to window rows [jlo, jhi].  False when empty.  The range only has to be a
// superset (coverage is decided per cell in binary64), so the window coordinates come from the hardware
// reciprocal: X * rcp(W) is within 2^-22 |X / W| of the quotient, at most 2^-6 of a cell for coordinates up
// to 65536, and the half-cell offsets carry 2^-5 of slack.  W outside [2^-100, 2^100] (where the reciprocal
// or the product could leave the normal range) takes the correctly rounded division.
__device__ __forceinline__ bool cell_range(const float (&X)[3], const float (&Y)[3], const float (&W)[3], int cols,
                                           int jlo, int jhi, int &i0, int &i1, int &j0, int &j1) {
    const float wmin = fminf(fminf(W[0], W[1]), W[2]), wmax = fmaxf(fmaxf(W[0], W[1]), W[2]);
    float sx[3], sy[3];
    if (wmin >= 0x1p-100f && wmax <= 0x1p100f) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float r = __builtin_amdgcn_rcpf(W[k]);
            sx[k] = X[k] * r; sy[k] = Y[k] * r;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) { sx[k] = X[k] / W[k]; sy[k] = Y[k] / W[k]; }
    }
    const float minx = fminf(fminf(sx[0], sx[1]), sx[2]), maxx = fmaxf(fmaxf(sx[0], sx[1]), sx[2]);
    const float miny = fminf(fminf(sy[0], sy[1]), sy[2]), maxy = fmaxf(fmaxf(sy[0], sy[1]), sy[2]);
    const float lox = ceilf(fmaxf(minx - 0.53125f, 0.0f)), hix = floorf(fminf(maxx - 0.46875f, float(cols - 1)));
    const float loy = ceilf(fmaxf(miny - 0.53125f, float(jlo))), hiy = floorf(fminf(maxy - 0.46875f, float(jhi)));
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

}  // namespace
