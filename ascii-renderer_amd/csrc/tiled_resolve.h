// tiled_resolve.h - tiled path, second half: per-tile LDS resolve and shading (k_tile)
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

// Colour of cell (i, j) for a sub-triangle the near plane cut.  sc holds the sub-triangle's screen
// vertices, so only the varyings of its three vertices are missing: each is a mesh vertex or the point
// I + (O - I) * (dI / (dI - dO)) on the edge from an inside vertex I to an outside vertex O, exactly as
// clip_near builds them.  Which is which follows from the signs of d = z_c + w_c of the three mesh
// vertices; the varyings are then rebuilt and interpolated one component at a time, which keeps this
// rarely taken path from setting the register budget of the kernel that contains it.
template <bool TEX>
__device__ __forceinline__ void shade_clipped_record(const FrameParams &fp, const float *__restrict__ vbo,
                                                     const uint32_t *__restrict__ idx, const uint4 *__restrict__ triuv,
                                                     const uint32_t *__restrict__ atlas, const Screen &sc, uint32_t id,
                                                     uint32_t flags, int i, int j, uint8_t (&px)[4]) {
    const uint32_t t = id >> 1, sub = id & 1u;
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
    edges_from_screen(sc, e);
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
    if (fp.has_model) {
        // world position and normal of the three mesh vertices under the model matrix, one component at a time
        for (int q = 0; q < 6; q++) {
            const int r = q < 3 ? q : q - 3, o = q < 3 ? 0 : 3;
            float w[3];
            for (int k = 0; k < 3; k++) {
                w[k] = (fp.model[r] * vp[k][o] + fp.model[3 + r] * vp[k][o + 1]) + fp.model[6 + r] * vp[k][o + 2];
                if (q < 3) w[k] = w[k] + fp.model[9 + r];
            }
            v[q] = varying(w[0], w[1], w[2]);
        }
        for (int q = 6; q < 9; q++) v[q] = varying(vp[0][q], vp[1][q], vp[2][q]);
    } else {
        for (int q = 0; q < 9; q++) v[q] = varying(vp[0][q], vp[1][q], vp[2][q]);
    }
    if (TEX && (flags & kFlagTextured)) {
        const uint4 uv = triuv[t];
        const float tu = varying(float(uv.x & 0xFFFFu), float(uv.y & 0xFFFFu), float(uv.z & 0xFFFFu));
        const float tv = varying(float(uv.x >> 16), float(uv.y >> 16), float(uv.z >> 16));
        if (apply_texture(fp, atlas, tu, tv, v + 6, px)) return;
    }
    shade(fp, v, px);
}

// Colour of cell (i, j) from the winner's triangle record: its setup is redone (same expressions as
// k_setup, so the same values), then a flat triangle takes normal and colour from the record itself, any other
// from its attribute row; world positions (point lights) are the record's.  Clipped sub-triangles go through
// shade_clipped_record.
template <bool TEX, typename FP>
__device__ __forceinline__ void shade_from_triangle(const FrameParams &fp, const FP &kp, const float *__restrict__ vbo,
                                                    const uint32_t *__restrict__ idx, const float *__restrict__ triattr,
                                                    const uint4 *__restrict__ triuv, const uint32_t *__restrict__ atlas,
                                                    const TriStatic &ts, uint32_t id, bool clipped, int i, int j,
                                                    uint8_t (&px)[4]) {
    Screen sc;
    if (clipped) {
        bool cut;
        binned_triangle_screen(fp, ts.p, id & 1u, sc, cut);
        shade_clipped_record<TEX>(fp, vbo, idx, triuv, atlas, sc, id, ts.flags, i, j, px);
        return;
    }
    {   // (clip-space z is not needed: the depth is the key's)
        const Vert<0> in0 = transform_position(kp, ts.p[0], ts.p[1], ts.p[2]), in1 = transform_position(kp, ts.p[3], ts.p[4], ts.p[5]),
                      in2 = transform_position(kp, ts.p[6], ts.p[7], ts.p[8]);
        to_screen_valid(in0.c, in1.c, in2.c, kp.hw, kp.hh, sc);
    }
    const uint32_t t = id >> 1;
    Edges e;
    edges_from_screen(sc, e);
    const double pxd = double(i) + 0.5, pyd = double(j) + 0.5;  // (the cell is covered and its depth is the key's: no test here)
    float f[3];
#pragma unroll
    for (int k = 0; k < 3; k++) f[k] = float(e.a[k] * pxd + (e.b[k] * pyd + e.c[k]));
    const float rS = 1.0f / ((f[0] + f[1]) + f[2]);
    const float b0 = f[0] * rS, b1 = f[1] * rS, b2 = f[2] * rS;
    float v[9];
    if (ts.flags & kFlagFlat) {
        // the same three-term sum as for a smooth triangle, with one value at all three vertices
        float n[3] = {ts.n[0], ts.n[1], ts.n[2]};
        if (kp.has_model) model_normal(kp, ts.n[0], ts.n[1], ts.n[2], n);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            v[3 + q] = (b0 * n[q] + b1 * n[q]) + b2 * n[q];
            v[6 + q] = (b0 * ts.c[q] + b1 * ts.c[q]) + b2 * ts.c[q];
        }
    } else {
        const float4 *row = reinterpret_cast<const float4 *>(triattr) + size_t(t) * 6;
        const float4 a0 = row[0], a1 = row[2], a2 = row[4];
        const float2 c0 = *reinterpret_cast<const float2 *>(row + 1), c1 = *reinterpret_cast<const float2 *>(row + 3),
                     c2 = *reinterpret_cast<const float2 *>(row + 5);
        float n0[3] = {a0.x, a0.y, a0.z}, n1[3] = {a1.x, a1.y, a1.z}, n2[3] = {a2.x, a2.y, a2.z};
        if (kp.has_model) {
            model_normal(kp, a0.x, a0.y, a0.z, n0);
            model_normal(kp, a1.x, a1.y, a1.z, n1);
            model_normal(kp, a2.x, a2.y, a2.z, n2);
        }
#pragma unroll
        for (int q = 0; q < 3; q++) v[3 + q] = (b0 * n0[q] + b1 * n1[q]) + b2 * n2[q];
        v[6] = (b0 * a0.w + b1 * a1.w) + b2 * a2.w;
        v[7] = (b0 * c0.x + b1 * c1.x) + b2 * c2.x;
        v[8] = (b0 * c0.y + b1 * c1.y) + b2 * c2.y;
    }
    if (kp.n_pl > 0) {
        if (kp.has_model) {
            float w0[3], w1[3], w2[3];
            model_point(kp, ts.p[0], ts.p[1], ts.p[2], w0);
            model_point(kp, ts.p[3], ts.p[4], ts.p[5], w1);
            model_point(kp, ts.p[6], ts.p[7], ts.p[8], w2);
#pragma unroll
            for (int q = 0; q < 3; q++) v[q] = (b0 * w0[q] + b1 * w1[q]) + b2 * w2[q];
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) v[q] = (b0 * ts.p[q] + b1 * ts.p[3 + q]) + b2 * ts.p[6 + q];
        }
    } else {
        v[0] = v[1] = v[2] = 0.0f;  // only the point-light loop reads the position
    }
    if (TEX && (ts.flags & kFlagTextured)) {
        const uint4 uv = triuv[t];
        const float tu = (b0 * float(uv.x & 0xFFFFu) + b1 * float(uv.y & 0xFFFFu)) + b2 * float(uv.z & 0xFFFFu);
        const float tv = (b0 * float(uv.x >> 16) + b1 * float(uv.y >> 16)) + b2 * float(uv.z >> 16);
        if (apply_texture(fp, atlas, tu, tv, v + 6, px)) return;
    }
    shade(kp, v, px);
}

// Inclusive prefix sum over the 64 lanes of a wavefront with data-parallel-primitive moves: a Hillis-Steele scan
// inside each row of 16 lanes (shifts by 1, 2, 4, 8; lanes without a source add 0), then the last lane of row 0 / 2
// is added to row 1 / 3, then lane 31 (the sum of rows 0 and 1) to rows 2 and 3.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
    int v = int(x);
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return uint32_t(v);
}

// Scans rows [r0, r1] x columns [i0, i1] of one sub-triangle into the LDS key tile; the rows times the width
// rounded up to a power of two must not exceed 64.
// Most cells of a cell range lie outside the triangle, and lanes of a wavefront walk ranges of very
// different sizes, so the scan is split in two: (1) per row, each edge function is evaluated in
// binary32 and solved for the column where it crosses a conservative error band, which gives the
// interval of columns that are not certainly outside - a run of bits in a 64-cell candidate mask (row r gets the
// 2^ls bits from r << ls, 2^ls the next power of two >= the width);
// (2) only those cells pay for the binary64 evaluation that decides coverage and feeds the depth.
// The result is exactly that of testing every cell in binary64.
// Band: with a32 = fl(a), r32 = fl(b py + c), |a32 px + r32 - E| <= 2^-24 (|a| px + |r|) (1 + eps); a cell
// with E >= 0 therefore has a32 px + r32 >= -band / 32 for band = 2^-18 (|a32| px_max + |r32|), and lies
// beyond the crossing column x of "a32 px + r32 = -band" by at least 0.8 band / |a32| >= 2^-19 |x|, far more
// than the 2^-22 |x| the binary32 evaluation of x can be off.  Edges with a32 == 0 (or an overflowing
// reciprocal) put no bound on the interval.
// Registers are what limits the kernel around this function, so each pass holds only what it needs: pass (1) the
// row coefficients b, c in binary64 and a in binary32 with its reciprocal; pass (2) a, b, c in binary64.  a is
// therefore formed twice from the same operands (the same value both times: one rounding of an exact difference),
// the second time behind a barrier the optimiser cannot see through - it would otherwise keep the first.
__device__ __forceinline__ void raster_small(const FrameParams &fp, Screen sc, uint32_t key_low, int i0, int i1,
                                             int r0, int r1, int tx0, int tr0, uint64_t *keys) {
    constexpr float kBand = 1.0f / 262144.0f;  // 2^-18
    const int w = i1 - i0 + 1;
    const int ls = w <= 1 ? 0 : 32 - __clz(w - 1);
    double b[3], c[3];
    unsigned long long cand = 0ull;
    {
        float a32[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int p = (k + 1) % 3, q = (k + 2) % 3;
            a32[k] = float(diff_of_products(sc.Y[p], sc.W[q], sc.Y[q], sc.W[p]));
            b[k] = diff_of_products(sc.X[q], sc.W[p], sc.X[p], sc.W[q]);
            c[k] = diff_of_products(sc.X[p], sc.Y[q], sc.X[q], sc.Y[p]);
        }
        // (hardware reciprocal: 1 ulp, inside the error budget above)
        const float inv_a[3] = {__builtin_amdgcn_rcpf(a32[0]), __builtin_amdgcn_rcpf(a32[1]), __builtin_amdgcn_rcpf(a32[2])};
        // (the band's |a32| px_max term, already times 2^-18: a power of two, so scaling first changes nothing)
        const float apx[3] = {kBand * (fabsf(a32[0]) * float(i1 + 1)), kBand * (fabsf(a32[1]) * float(i1 + 1)),
                              kBand * (fabsf(a32[2]) * float(i1 + 1))};
        const float flo = float(i0), fhi = float(i1);
        for (int rr = r0; rr <= r1; rr++) {
            const double py = double(fp.rows - 1 - rr) + 0.5;
            float lo = flo, hi = fhi;  // candidate columns, as cell indices
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float r32 = float(__builtin_fma(b[k], py, c[k]));  // (fused: closer to b py + c than the bound assumes)
                const float band = __builtin_fmaf(kBand, fabsf(r32), apx[k]);
                const float x = __builtin_fmaf(-band - r32, inv_a[k], -0.5f);  // crossing column as a cell index (fused: nearer the true crossing)
                lo = fmaxf(lo, a32[k] > 0.0f ? x : flo);
                hi = fminf(hi, a32[k] < 0.0f ? x : fhi);
            }
            if (!(lo <= hi)) continue;
            const int ilo = int(ceilf(lo)), ihi = int(floorf(hi));
            const int n = ihi - ilo + 1;
            if (n <= 0) continue;
            const unsigned long long run = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
            cand |= run << (((rr - r0) << ls) + (ilo - i0));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) asm volatile("" : "+v"(sc.Y[k]), "+v"(sc.W[k]));
    // coverage is E_k >= thr_k with thr_k = 0 on edges that own their boundary and the smallest positive
    // binary64 otherwise (E_k > 0): the same decision as edge_inside with one comparison per edge
    double a[3], thr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        a[k] = diff_of_products(sc.Y[p], sc.W[q], sc.Y[q], sc.W[p]);
        thr[k] = (a[k] > 0.0 || (a[k] == 0.0 && b[k] > 0.0)) ? 0.0 : 4.9406564584124654e-324;
    }
    uint32_t clo = uint32_t(cand), chi = uint32_t(cand >> 32);
    while (clo | chi) {
        int k;
        if (clo) { k = __ffs(int(clo)) - 1; clo &= clo - 1u; }
        else { k = 31 + __ffs(int(chi)); chi &= chi - 1u; }
        const int rr = r0 + (k >> ls), i = i0 + (k & ((1 << ls) - 1));
        const double px = double(i) + 0.5, py = double(fp.rows - 1 - rr) + 0.5;
        const double E0 = a[0] * px + (b[0] * py + c[0]);
        const double E1 = a[1] * px + (b[1] * py + c[1]);
        const double E2 = a[2] * px + (b[2] * py + c[2]);
        const float f0 = float(E0), f1 = float(E1), f2 = float(E2);
        const float den = (f0 * sc.W[0] + f1 * sc.W[1]) + f2 * sc.W[2];
        const float num = (f0 * sc.Z[0] + f1 * sc.Z[1]) + f2 * sc.Z[2];
        const float zw = (num / den) * 0.5f + 0.5f;
        if (int(E0 >= thr[0]) & int(E1 >= thr[1]) & int(E2 >= thr[2]) & int(zw >= 0.0f) & int(zw < 1.0f))
            atomicMin(reinterpret_cast<unsigned long long *>(&keys[(rr - tr0) * kTileW + (i - tx0)]),
                      (unsigned long long)make_key(zw, key_low));
    }
}

// Where the entry with place p in the order of work classes sits in a tile's list: every second run of 256 places is
// stored back to front.  The lanes of the workgroup take the list 256 places at a time, first wavefront first; with the
// list in ascending order of work that would give the lightest quarter of every run to the first wavefront and the
// heaviest to the last, this way light and heavy quarters alternate.  (Its own inverse.  A list that ends inside a
// reversed run leaves the head of that run empty: slot s holds an entry if and only if list_slot(s) < length.)
__device__ __forceinline__ uint32_t list_slot(uint32_t p) { return p ^ ((0u - ((p >> 8) & 1u)) & 255u); }

// TEX: the frame samples an atlas (texel coordinates and atlas are only read then).  SH: log2 of the tile height.
// BATCH: one launch for several frames (parameters from fps, per-frame buffers at the frame's stride).  The workgroups of
// one tile in the launch's frames are neighbours in the launch order and land on the same XCD (w = ((b / 8) * frames + f)
// * 8 + b % 8 for tile workgroup b): consecutive frames of a moving camera list mostly the same triangles in a tile, so the
// records one frame's tile pulls into that XCD's L2 serve the other frames' as well.
template <bool TEX, int SH, bool BATCH>
__global__ __launch_bounds__(kTileThreads) void k_tile(FrameParams fp_one, const float *__restrict__ vbo,
                                              const uint32_t *__restrict__ idx, const float *__restrict__ triattr,
                                              const uint4 *__restrict__ triuv, const uint32_t *__restrict__ atlas,
                                              const TriStatic *__restrict__ tri, const TriStatic *__restrict__ tri_t,
                                              const uint32_t *__restrict__ entries, uint32_t entries_cap,
                                              const BinState *__restrict__ state, uint32_t *__restrict__ blocksum,
                                              uint64_t *__restrict__ vis,
                                              uint8_t *__restrict__ rgba, float *__restrict__ depth,
                                              uint8_t *__restrict__ ascii, uint8_t *__restrict__ ridx,
                                              uint32_t *__restrict__ ids, const FrameParams *__restrict__ fps, BatchStrides bs) {
    constexpr int kTileH = 1 << SH, kTileCells = kTileW * kTileH;
    uint32_t wg = blockIdx.x, frm = 0;  // the tile's workgroup number within its frame, the frame
    if (BATCH) {
        const uint32_t q = blockIdx.x >> 3;
        frm = q % bs.frames;
        wg = (q / bs.frames) * 8u + (blockIdx.x & 7u);
    }
    const FrameParams &fp = BATCH ? fps[frm] : fp_one;
    if (BATCH) {
        const size_t f = frm;
        entries += f * bs.entries; blocksum += f * bs.dircount; vis += f * bs.cells; rgba += f * bs.cells * 4; depth += f * bs.cells;
        ascii += f * bs.ascii; ridx += f * bs.cells;
        if (ids) ids += f * bs.cells;
    }
    __shared__ uint64_t keys[kTileCells];
    __shared__ uint32_t list[kTileListMax];
    __shared__ uint32_t big[kTileBigMax];
    // covered cells for the shading step; before that, staging for kTileStage swept records of 64 bytes
    __shared__ __align__(16) uint16_t covered[kTileCells > kTileStage * 32 ? kTileCells : kTileStage * 32];
    __shared__ uint32_t n_big, n_cov, n_marked;
    __shared__ uint32_t cls_count[2][64];  // entries per work class in this filter step (two sets, used in turn)
    const uint32_t tid = threadIdx.x;
    // Workgroups are handed to the eight XCDs round-robin and each XCD has its own L2: the tiles of one
    // bin, which all read the bin's entry array, are given workgroup numbers that are equal modulo 8.
    // Workgroup B = 8 * tiles_per_bin * g + 8 * k + x is tile k of bin 8 g + x.
    // (a frame without k_binscan: k_fill has used the totals per bin and row block that k_setup added up - they go
    // back to zero for the next such frame on this set of buffers)
    if (fp.fold_scan && wg == 0)
        for (int k = int(threadIdx.x); k < 16 * fp.nbx * fp.nby; k += kTileThreads) blocksum[k] = 0;
    const int side = 1 << fp.bshift, tiles_per_bin = side * side;
    const int wg_group = int(wg) / (8 * tiles_per_bin), wg_in = int(wg) % (8 * tiles_per_bin);
    const int bin = wg_group * 8 + (wg_in & 7), sub = wg_in >> 3;
    if (bin >= fp.nbx * fp.nby) return;
    const int tile_x = (bin % fp.nbx) * side + (sub & (side - 1)), tile_y = (bin / fp.nbx) * side + (sub >> fp.bshift);
    if (tile_x >= fp.ntx || tile_y >= fp.nty) return;
    const int tx0 = tile_x * kTileW, tr0 = fp.ya + tile_y * kTileH;                      // first column / output row
    const int tx1 = min(tx0 + kTileW, fp.cols) - 1, tr1 = min(tr0 + kTileH, fp.yb) - 1;  // last column / row
    const uint32_t bit = 1u << sub;
    const int jlo = fp.rows - fp.yb, jhi = fp.rows - 1 - fp.ya;  // window rows of the frame's strip, as in k_setup
    const BinState &st = state[fp.set];
    for (int c = int(tid); c < kTileCells; c += kTileThreads) keys[c] = kBackgroundKey;
    if (tid == 0) { n_big = 0; n_cov = 0; n_marked = 0; }
    if (tid < 128u) cls_count[tid >> 6][tid & 63u] = 0;
    const uint32_t e_begin = st.start[bin];
    const uint32_t e_total = st.total[bin];
    const bool stored = e_begin <= entries_cap && e_total <= entries_cap - e_begin;  // same test as k_fill
    const uint32_t e_end = stored ? e_begin + e_total : e_begin;
    __syncthreads();

    // filter the bin's entries by this tile's mask bit into the LDS list; whenever the list could
    // overflow on the next step (or at the end) rasterise what it holds
    // (kFilterBatch entries per lane and step, all loads in flight together: the step is one memory latency)
    uint32_t nl = 0;  // length of the list (the same in every lane)
    int cset = 0;
    for (uint32_t eb = e_begin;; eb += kTileThreads * kFilterBatch) {
        uint32_t ent[kFilterBatch];
#pragma unroll
        for (int q = 0; q < kFilterBatch; q++) {
            const uint32_t e = eb + uint32_t(q) * uint32_t(kTileThreads) + tid;
            ent[q] = e < e_end ? entries[e] : 0u;  // mask 0 never matches
        }
        // The step's matches go to the list ordered by work class (a counting sort: rank within the class from an
        // LDS counter, start of the class from a scan over the counters), so that the 64 consecutive entries a
        // wavefront takes below are triangles with the same number of rows and about the same width.  Every
        // wavefront scans the counters for itself (lane c: class c), which keeps the list length in a register -
        // the same in all lanes of the workgroup, as the flush decision below has to be: its block contains barriers.
        const uint32_t id_mask = (1u << fp.id_bits) - 1u, cls_mask = (1u << fp.cls_bits) - 1u;
        uint32_t rank[kFilterBatch];
#pragma unroll
        for (int q = 0; q < kFilterBatch; q++)
            if ((ent[q] >> kEntryMaskShift) & bit) rank[q] = atomicAdd(&cls_count[cset][(ent[q] >> fp.id_bits) & cls_mask], 1u);
        __syncthreads();
        {
            const uint32_t v = cls_count[cset][tid & 63u];
            const uint32_t inc = wave_inclusive_scan(v);
            const uint32_t start = nl + inc - v;  // where this lane's class begins in the list
#pragma unroll
            for (int q = 0; q < kFilterBatch; q++) {
                const uint32_t base = uint32_t(__shfl(int(start), int((ent[q] >> fp.id_bits) & cls_mask)));  // (all lanes take part)
                if ((ent[q] >> kEntryMaskShift) & bit) list[list_slot(base + rank[q])] = ent[q] & id_mask;
            }
            nl += uint32_t(__builtin_amdgcn_readlane(int(inc), 63));
            // the other set was last read before the previous step's closing barrier and is next used after this one's
            if (tid < 64u) cls_count[cset ^ 1][tid] = 0;
            cset ^= 1;
        }
        __syncthreads();
        const bool last = eb + kTileThreads * kFilterBatch >= e_end;
        if (nl + kTileThreads * kFilterBatch > uint32_t(kTileListMax) || last) {
            // sub-triangles with a small cell range inside the tile: one lane each; others are marked in the list for
            // the sweep below (a lane's next triangle record is fetched while it works on the current one).
            // A short list leaves most lanes without a triangle, and the few that have one walk their rows one
            // after the other: then 2, 4, 8 or 16 neighbouring lanes share a triangle - each redoes its setup (for
            // nothing: the wavefront executes those instructions anyway) and takes an equal share of its rows.
            int gshift = 0;
            while (gshift < 4 && (nl << (gshift + 1)) <= uint32_t(kTileThreads)) gshift++;
            const uint32_t per_round = uint32_t(kTileThreads) >> gshift;  // groups of sharing lanes: entries taken per round
            uint32_t first = tid >> gshift;
            uint32_t my_part = tid & ((1u << gshift) - 1u);
            const uint32_t n_rounds = (nl + per_round - 1u) / per_round;
            // One round and groups to spare (a list between half and all of what a round takes): the spare groups join
            // the last entries of the list - the heaviest, the list being in order of work - whose rows are then split
            // among twice as many lanes.  Entry e gets the spare group 2 nl - 1 - e.
            uint32_t spare_from = ~0u;  // entries from here on have two groups (none in a list of several rounds)
            if (n_rounds == 1u) {
                spare_from = 2u * nl > per_round ? 2u * nl - per_round : 0u;
                if (first >= nl) {
                    first = (2u * nl - 1u - first) & 0xFFFFu;  // (beyond 2 nl - 1 this wraps to a place past the list: no entry)
                    my_part += 1u << gshift;
                }
            }
            // A list word may carry kListSweep by the time it is read here: with spare groups two lane groups - usually in
            // different wavefronts - work on the same entry, and part 0 marks it below without a barrier in between.  Every
            // read in this loop therefore masks the mark off (with it the word would index 2^30 records past the array).
            TriStatic ahead;
            if (first < nl) ahead = tri[(list[first] & ~kListSweep) >> 1];  // (the first run of 256 is in order)
            // (list place and the lane's number among the lanes of its entry travel in one register: they are few bits each,
            // and registers are what limits this kernel)
            for (uint32_t rp = first | (my_part << 16); (rp & 0xFFFFu) < n_rounds * per_round; rp += per_round) {
                const uint32_t r = rp & 0xFFFFu;
                const int part = int(rp >> 16);
                if (list_slot(r) >= nl) continue;  // (an empty slot: only in the last round)
                const uint32_t id = list[r] & ~kListSweep;
                const TriStatic ts = ahead;
                if (list_slot(r + per_round) < nl) ahead = tri[(list[r + per_round] & ~kListSweep) >> 1];
                Screen sc;
                bool clipped;
                int ci0, ci1, cj0, cj1;
                binned_triangle_screen(fp, ts.p, id & 1u, sc, clipped);
                if (!cell_range(sc.X, sc.Y, sc.W, fp.cols, jlo, jhi, ci0, ci1, cj0, cj1)) continue;  // (binned: never empty)
                const int i0 = max(ci0, tx0), i1 = min(ci1, tx1);
                int r0 = max(fp.rows - 1 - cj1, tr0), r1 = min(fp.rows - 1 - cj0, tr1);
                // a lane's share of the rows, times the width rounded up to a power of two, has to fit the 64-bit
                // candidate mask of raster_small
                const int pshift = gshift + (r >= spare_from ? 1 : 0);  // log2 of the lanes that share this entry
                const int len = (r1 - r0 + (1 << pshift)) >> pshift;
                const int w = i1 - i0 + 1;
                if ((len << (w <= 1 ? 0 : 32 - __clz(w - 1))) > kTileSmallCells) {
                    if (part == 0) { list[r] = id | kListSweep; n_marked = 1u; }  // (the only writer of this word; readers mask the mark off)
                    continue;
                }
                r0 += part * len;  // this lane's share of the rows
                r1 = min(r0 + len - 1, r1);
                if (r0 > r1) continue;
                const uint32_t key_id = 2u * (ts.flags >> kFlagBits) + (id & 1u);
                raster_small(fp, sc, tile_key_low(key_id, clipped), i0, i1, r0, r1, tx0, tr0, keys);
            }
            __syncthreads();
            // sweep: the marked entries are collected 256 list places at a time; when more than half of the sweep list
            // is taken (or at the end) their triangle records are brought into LDS 64 at a time (one fetch latency per
            // batch instead of one per triangle), and every wavefront sweeps a triangle of its own, 64 cells per step
            TriStatic *stage = reinterpret_cast<TriStatic *>(covered);  // (the cell list is not in use before the raster step ends)
            const uint32_t sweep_end = n_marked != 0u ? nl : 0u;  // (nothing marked - the usual case at a million small triangles: no pass over the list)
            for (uint32_t s0 = 0; s0 < sweep_end; s0 += kTileThreads) {
                const uint32_t marked = list_slot(s0 + tid) < nl ? list[s0 + tid] : 0u;
                if (marked & kListSweep) big[atomicAdd(&n_big, 1u)] = marked & ~kListSweep;
                __syncthreads();
                const uint32_t nb = n_big;  // (at most kTileBigMax: it was at most half of that before this step)
                __syncthreads();            // (every lane has read it before it changes again)
                if (tid == 0) n_marked = 0;  // (read by all before the barriers above; next written after the barriers below or the next step's)
                if (nb == 0u || (nb <= uint32_t(kTileBigMax - kTileThreads) && s0 + kTileThreads < nl)) continue;
                for (uint32_t b0 = 0; b0 < nb; b0 += kTileStage) {
                    const uint32_t ns = min(uint32_t(kTileStage), nb - b0);
                    if (tid < ns) stage[tid] = tri[big[b0 + tid] >> 1];
                    __syncthreads();
                    for (uint32_t q = tid >> 6; q < ns; q += kTileThreads / 64) {
                        const uint32_t id = big[b0 + q];
                        const uint32_t key_id = 2u * (stage[q].flags >> kFlagBits) + (id & 1u);
                        float p[9];
#pragma unroll
                        for (int k = 0; k < 9; k++) p[k] = stage[q].p[k];
                        Screen sc;
                        bool clipped;
                        int ci0, ci1, cj0, cj1;
                        binned_triangle_screen(fp, p, id & 1u, sc, clipped);
                        if (!cell_range(sc.X, sc.Y, sc.W, fp.cols, jlo, jhi, ci0, ci1, cj0, cj1)) continue;
                        const int i0 = max(ci0, tx0), i1 = min(ci1, tx1);
                        const int r0 = max(fp.rows - 1 - cj1, tr0), r1 = min(fp.rows - 1 - cj0, tr1);
                        const int w = i1 - i0 + 1, n = w * (r1 - r0 + 1);
                        Edges ed;
                        edges_from_screen(sc, ed);
                        for (int k = int(tid & 63u); k < n; k += 64) {
                            const int dr = k / w, i = i0 + (k - dr * w), rr = r0 + dr;
                            float f[3], zw;
                            if (cover(ed, i, fp.rows - 1 - rr, f, zw))
                                atomicMin(reinterpret_cast<unsigned long long *>(&keys[(rr - tr0) * kTileW + (i - tx0)]),
                                          (unsigned long long)make_key(zw, tile_key_low(key_id, clipped)));
                        }
                    }
                    __syncthreads();
                }
                if (tid == 0) n_big = 0;
                __syncthreads();
            }
            nl = 0;
        }
        if (last) break;
    }

    // sub-triangles k_fill could not store were resolved through the global key buffer
    if (st.overflow != 0) {
        for (int c = int(tid); c < kTileCells; c += kTileThreads) {
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
    for (int c = int(tid); c < kTileCells; c += kTileThreads)
        if (keys[c] != kBackgroundKey) covered[atomicAdd(&n_cov, 1u)] = uint16_t(c);
    __syncthreads();
    const uint32_t nc = n_cov;
    for (uint32_t k = tid; k < nc; k += kTileThreads) {
        const int c = int(covered[k]);
        const int rr = tr0 + (c >> 6), i = tx0 + (c & 63);
        const uint32_t low = uint32_t(keys[c]);  // the key's low word is the winner: (2*triangle + sub) << 1 | clipped
        const uint32_t id = low >> 1;
        const TriStatic ts = tri_t[id >> 1];  // (the winner is known by its own index: the copy kept in that order)
        uint8_t px[4];
        if constexpr (BATCH) shade_from_triangle<TEX>(fp, fp, vbo, idx, triattr, triuv, atlas, ts, id, (low & 1u) != 0u, i, fp.rows - 1 - rr, px);
        else shade_from_triangle<TEX>(fp, *kernarg_params(), vbo, idx, triattr, triuv, atlas, ts, id, (low & 1u) != 0u, i, fp.rows - 1 - rr, px);
        if (fp.want_ids) ids[size_t(rr) * fp.cols + i] = id;
        // untextured frames have alpha 255 everywhere: the byte carries the cell's ramp index to the output step
        // instead, which then has no arithmetic left to do per cell
        const uint32_t top = TEX ? uint32_t(px[3]) : uint32_t(ramp_index(int(px[0]), int(px[1]), int(px[2]), fp.ramp_len));
        keys[c] = (keys[c] & 0xFFFFFFFF00000000ull) |  // (the depth the raster pass computed for this winner stays)
                  (uint32_t(px[0]) | (uint32_t(px[1]) << 8) | (uint32_t(px[2]) << 16) | (top << 24));
    }
    __syncthreads();

    // colour and ramp index of a resolved key (a shaded cell always has depth < 1.0 in the high word)
    auto decode = [&](uint64_t k, uint32_t &col, int &ri) {
        const bool bg = k == kBackgroundKey;
        if (TEX) {
            col = bg ? 0xFF000000u : uint32_t(k);
            ri = ramp_index(int(col & 255u), int((col >> 8) & 255u), int((col >> 16) & 255u), fp.ramp_len);
        } else {
            col = bg ? 0xFF000000u : (uint32_t(k) | 0xFF000000u);
            ri = bg ? 0 : int(uint32_t(k) >> 24);  // (black is ramp index 0)
        }
        return bg;
    };
    // the tile's rows.  Usual case: a lane writes four consecutive cells of a row, one 16-byte store each for colour
    // and depth and one dword of ramp indices (rows and the tile's columns start at multiples of four cells)
    if ((fp.cols & 3) == 0 && fp.mode_on && !fp.want_ids) {
        for (int q = int(tid); q < kTileCells / 4; q += kTileThreads) {
            const int row = q >> 4, x = (q & 15) * 4;
            const int rr = tr0 + row, i = tx0 + x;
            if (i > tx1 || rr > tr1) continue;  // (a group of four lies inside or outside the grid as a whole)
            const size_t cell = size_t(rr) * fp.cols + i;
            uint32_t col[4], ri4 = 0;
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t key = keys[row * kTileW + x + k];
                int ri;
                const bool bg = decode(key, col[k], ri);
                d[k] = bg ? 1.0f : __uint_as_float(uint32_t(key >> 32));
                ri4 |= uint32_t(ri) << (8 * k);
            }
            reinterpret_cast<uint4 *>(rgba)[cell >> 2] = make_uint4(col[0], col[1], col[2], col[3]);
            reinterpret_cast<float4 *>(depth)[cell >> 2] = make_float4(d[0], d[1], d[2], d[3]);
            reinterpret_cast<uint32_t *>(ridx)[cell >> 2] = ri4;
        }
        return;
    }
    for (int c = int(tid); c < kTileCells; c += kTileThreads) {
        const int rr = tr0 + (c >> 6), i = tx0 + (c & 63);
        if (i > tx1 || rr > tr1) continue;
        const size_t cell = size_t(rr) * fp.cols + i;
        const uint64_t k = keys[c];
        uint32_t col;
        int ri;
        const bool bg = decode(k, col, ri);
        reinterpret_cast<uint32_t *>(rgba)[cell] = col;
        depth[cell] = bg ? 1.0f : __uint_as_float(uint32_t(k >> 32));
        if (fp.want_ids && bg) ids[cell] = 0xFFFFFFFFu;
        if (fp.mode_on) ridx[cell] = uint8_t(ri);
        else if (rr >= fp.row0 && rr < fp.row1) {
            const uint32_t qa = col >> 24;  // 255 unless an atlas texel put an ASCII override there
            ascii[cell] = (TEX && qa >= 2u && qa <= 254u) ? uint8_t(qa) : fp.ramp[ri];
        }
    }
}

}  // namespace
