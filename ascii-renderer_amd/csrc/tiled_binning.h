// tiled_binning.h - tiled path, first half: mesh streams, triangle setup, bin scans, entry fill
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// Tiled path: sub-triangles are binned to screen bins of 2x2 tiles, each tile - 64 cells wide, 32 rows high, or
// 16 or 8 when the frame (a strip, a small grid) would otherwise have too few tiles to fill the chip - is
// resolved in LDS.
//
//   k_tristream (at mesh upload) gathers, per triangle, the three positions into nine planes of
//             n_tris floats (36 contiguous, coalesced bytes per triangle for the per-frame pass),
//             the three vertices' normals and colours into one 96-byte row (one gather per shaded
//             cell of a smooth triangle instead of index -> vertex chasing) and one 64-byte triangle
//             record (positions, flat normal and colour, flags) that is all a tile needs of a flat
//             triangle.  None of this changes from frame to frame.
//   k_setup   one pass over the position planes, one triangle per lane and pass: transform, near
//             clip, cull, cell range.  Every sub-triangle that can produce fragments leaves an
//             8-byte tuple (cell range) and its id 2*triangle + sub in the workgroup's block of
//             survivor slots (dense, in triangle order), and is counted into the workgroup's LDS
//             histogram over bins, which is then published as one row of the count table.
//   k_binscan column-wise exclusive prefix of the count table over the k_setup workgroups (in
//             place, two levels) and per-row-block bin totals: every (workgroup, bin) pair gets a
//             fixed slice of the bin's entry array without a single global atomic.
//   k_fill    bin totals -> start of every bin's entry array (each workgroup scans them for itself), then
//             every k_setup workgroup's survivors become one entry (id, work class, mask of the bin's
//             tiles the cell range touches) per (survivor, bin), written into that workgroup's slice.
//   k_tile    one workgroup per tile: filters its bin's entries by mask bit into an LDS work list kept in
//             order of work class, fetches the listed triangles' records, redoes their setup (same
//             expressions as k_setup, hence the same values) and rasterises them into an LDS key tile
//             (64 x 32, 16 or 8 cells) with LDS atomic min (small cell ranges: one lane - or a few -
//             per triangle; large ones: staged in LDS, one wavefront sweeps each), then shades the
//             covered cells and writes the tile's RGBA /
//             depth / ramp-index rows.  Visibility keys never touch HBM, and no per-frame setup
//             result does either: a scattered 64-byte read costs the memory side a whole request
//             whether it fetches a setup record or the static triangle, and the static triangle
//             does not have to be written first.
// If the entry pool cannot hold a bin, the sub-triangles touching that bin are rasterised straight
// into a global key buffer by k_fill, which also raises the frame's overflow flag, and k_tile merges
// that buffer in.
// Everything a frame writes exists kSets times, so that consecutive frames can overlap: see
// enqueue_frame.
// ------------------------------------------------------------------------------------------------
constexpr int kSets = 3;               // frames in flight: every per-frame buffer exists kSets times
constexpr int kTileW = 64;              // a tile is one wavefront wide ...
constexpr int kTileShiftMax = 5, kTileShiftMin = 3;  // ... and 32, 16 or 8 rows high (FrameParams::tshift, chosen per frame)
constexpr int kMaxTiles = 8000;       // tiles per frame the tiled path accepts
constexpr int kMaxBins = 2048;        // >= kMaxTiles / 4 for any tile-grid shape the path accepts
constexpr int kSetupThreads = 256;    // k_setup / k_fill workgroup size
constexpr uint32_t kMaxGroups = 1024;  // k_setup workgroups (rows of the count table)
constexpr int kMaxPasses = 512;        // passes of 256 slots a k_setup workgroup makes at most (2^27 triangles / 1024 / 256)
constexpr int kTileBigMax = 512;      // per-tile list of triangles swept cooperatively
constexpr uint32_t kListSweep = 0x80000000u;  // mark in a tile's list: the entry is for the cooperative sweep (ids stay below 2^28)
constexpr int kTileStage = 64;        // of which so many are staged in LDS at a time (64 B each, in the cell list's 4 KB)
constexpr int kTileSmallCells = 64;   // cell-range area (inside the tile) one lane scans itself
constexpr int kTileListMax = 2048;    // ids staged in LDS between the filter and the raster step
constexpr int kEntryMaskShift = 28;   // a bin entry is (2*slot + sub) | work class << id_bits | (mask of the bin's 2x2 tiles) << 28
constexpr int kMaxClassBits = 6;      // work classes of a bin entry: up to 64 (fewer when the mesh needs the bits for its slots)
constexpr uint32_t kEntryLocMask = (1u << kEntryMaskShift) - 1u;
constexpr int kTileThreads = 256;     // k_tile workgroup size
constexpr int kFilterBatch = 4;       // bin entries a k_tile lane loads per filter step
constexpr uint32_t kFlagFlat = 2u;    // the triangle's three vertices carry identical normal and colour
constexpr uint32_t kFlagTextured = 4u;  // the triangle samples the atlas (set from its texel-coordinate row)

// One triangle as the per-tile resolve wants it, written once per mesh: one 64-byte line holds the three
// positions, the flat normal and colour (valid when kFlagFlat is set) and the flags.  A tile fetches the
// triangles binned to it by this record alone and redoes the few hundred operations of their setup, instead
// of reading a per-frame setup record that another kernel had to write first.
struct __align__(16) TriStatic {
    float p[9];      // positions of the three vertices
    float n[3];      // normal (flat triangles)
    float c[3];      // colour (flat triangles)
    uint32_t flags;  // kFlagFlat | kFlagTextured in the low kFlagBits bits, the triangle's index t above them
};
static_assert(sizeof(TriStatic) == 64, "triangle records are four 16-byte loads");

// Per-frame binning state; one block per buffer set.
struct BinState {
    uint32_t total[kMaxBins];
    uint32_t start[kMaxBins];   // exclusive scan of total (k_fill)
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

// block_exclusive_scan of h0 + h1 for lanes that each hold two flags: two ballots and their bit counts below
// the lane take the place of the shuffle ladder.
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_scan_flags(bool h0, bool h1, uint32_t *s_wave, uint32_t &total) {
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long b0 = __ballot(h0), b1 = __ballot(h1);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(uint32_t(b0 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(b0), 0u)) +
                           __builtin_amdgcn_mbcnt_hi(uint32_t(b1 >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(b1), 0u));
    if ((threadIdx.x & 63u) == 0u) s_wave[wave] = uint32_t(__popcll(b0)) + uint32_t(__popcll(b1));
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
    return off + below;
}

// ---- mesh streams in spatial order ------------------------------------------------------------------------
// The per-frame pass walks the triangles in the order of their position planes, 256 at a time, and a tile fetches
// the records of the triangles binned to it.  Both want triangles that are close in space to be close in memory:
// a block of 256 consecutive triangles then touches few bins (long runs in every bin's entry array, and a tight
// bounding box that lets whole blocks be skipped when they miss the view or the strip), and the records a tile
// fetches share cache lines.  So at upload the triangles are sorted by the 30-bit Morton code of their centroid
// inside the mesh's bounding box (a stable radix sort: equal codes keep submission order) and everything the
// binning reads is laid out in that order, "slot" s = position in the sorted order.  Nothing a caller can see
// depends on it: visibility keys, winner ids and the tie rule use the triangle's own index t, which every record
// carries, and the shading step reads a second copy of the records kept in t order.

constexpr int kFlagBits = 3;  // a record's flags word: kFlagFlat | kFlagTextured below, the triangle's index t above

// float -> unsigned with the same order (for atomic min / max on coordinates)
__device__ __forceinline__ uint32_t float_order(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_unorder(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// Centroid of triangle t (any finite triangle; others report false).
__device__ __forceinline__ bool triangle_centroid(const float *__restrict__ vbo, const uint32_t *__restrict__ idx, uint32_t n_verts,
                                                  uint32_t t, float (&c)[3]) {
    c[0] = c[1] = c[2] = 0.0f;
    for (int v = 0; v < 3; v++) {
        const uint32_t vi = idx[3 * size_t(t) + v];
        if (vi >= n_verts) return false;
        const float *p = vbo + size_t(vi) * 9;
        for (int ax = 0; ax < 3; ax++) c[ax] += p[ax];
    }
    for (int ax = 0; ax < 3; ax++) c[ax] *= (1.0f / 3.0f);
    return isfinite(c[0]) && isfinite(c[1]) && isfinite(c[2]);
}

// bounds[0..2] = min, [3..5] = max of the centroids, as float_order keys (initialised to ~0 / 0 by the caller).
__global__ __launch_bounds__(256) void k_mesh_bounds(const float *__restrict__ vbo, const uint32_t *__restrict__ idx,
                                                     uint32_t n_verts, uint32_t n_tris, uint32_t *__restrict__ bounds) {
    uint32_t lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n_tris; t += gridDim.x * 256u) {
        float c[3];
        if (!triangle_centroid(vbo, idx, n_verts, t, c)) continue;
        for (int ax = 0; ax < 3; ax++) {
            const uint32_t k = float_order(c[ax]);
            lo[ax] = min(lo[ax], k); hi[ax] = max(hi[ax], k);
        }
    }
    for (int ax = 0; ax < 3; ax++) {
        for (int d = 32; d >= 1; d >>= 1) {
            lo[ax] = min(lo[ax], uint32_t(__shfl_xor(int(lo[ax]), d)));
            hi[ax] = max(hi[ax], uint32_t(__shfl_xor(int(hi[ax]), d)));
        }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(&bounds[ax], lo[ax]); atomicMax(&bounds[3 + ax], hi[ax]); }
    }
}

__device__ __forceinline__ uint32_t spread_bits_10(uint32_t v) {  // abcdefghij -> a00b00c00...j
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// code[t] = Morton code of triangle t's centroid in the box of k_mesh_bounds (0 for triangles without one), order[t] = t.
__global__ __launch_bounds__(256) void k_morton(const float *__restrict__ vbo, const uint32_t *__restrict__ idx, uint32_t n_verts,
                                                uint32_t n_tris, const uint32_t *__restrict__ bounds, uint32_t *__restrict__ code,
                                                uint32_t *__restrict__ order) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tris) return;
    float c[3];
    uint32_t m = 0;
    if (triangle_centroid(vbo, idx, n_verts, t, c)) {
        uint32_t q[3];
        for (int ax = 0; ax < 3; ax++) {
            const float lo = float_unorder(bounds[ax]), hi = float_unorder(bounds[3 + ax]);
            const float ext = hi - lo;
            const float u = (ext > 0.0f && isfinite(ext)) ? (c[ax] - lo) / ext : 0.0f;
            q[ax] = uint32_t(fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f));
        }
        m = spread_bits_10(q[0]) | (spread_bits_10(q[1]) << 1) | (spread_bits_10(q[2]) << 2);
    }
    code[t] = m;
    order[t] = t;
}

// Per slot s (perm[s] = t, the triangle in that place of the spatial order): positions -> planes pos[k*stride + s]
// (k = 3*vertex + axis); tri_s[s] = the triangle's 64-byte record with t in the upper bits of its flags word and
// kFlagFlat set when the three normals and the three colours are identical.  Per triangle t: the same record in
// tri_t[t] (the shading step's copy) and the normal and colour of each vertex -> attr[t*24 + 8*v ..] = (nx, ny,
// nz, r, g, b, 0, 0), so a vertex is one 16-byte plus one 8-byte load.  box[6 * (s / 256) ..] = bounding box
// (min xyz, max xyz) of the positions of each block of 256 slots (NaN coordinates left out).
// A triangle that names a vertex outside the buffer gets NaN positions, which every later stage rejects.
// ATTR = false (an update of positions alone, ar_update_positions): only what depends on the positions is rewritten -
// the planes, the nine position floats of both records, the boxes; normals, colours and flags stay as they are.
template <bool ATTR>
__global__ __launch_bounds__(256) void k_tristream(const float *__restrict__ vbo, const uint32_t *__restrict__ idx,
                                                   const uint32_t *__restrict__ perm, uint32_t n_verts, uint32_t n_tris,
                                                   size_t stride, float *__restrict__ pos, float *__restrict__ attr,
                                                   TriStatic *__restrict__ tri_s, TriStatic *__restrict__ tri_t,
                                                   float *__restrict__ box) {
    __shared__ float s_box[4][6];
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (s < n_tris) {
        const uint32_t t = perm[s];
        float a[3][9];
        bool all_ok = true, okv[3];
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const uint32_t vi = idx[3 * size_t(t) + v];
            const bool ok = vi < n_verts;
            okv[v] = ok;
            all_ok = all_ok && ok;
            const float *p = vbo + size_t(ok ? vi : 0u) * 9;
#pragma unroll
            for (int q = 0; q < (ATTR ? 9 : 3); q++) a[v][q] = ok ? p[q] : 0.0f;
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const float x = ok ? p[ax] : __builtin_nanf("");
                pos[size_t(3 * v + ax) * stride + s] = x;
                lo[ax] = fminf(lo[ax], x); hi[ax] = fmaxf(hi[ax], x);  // (fminf / fmaxf drop a NaN operand)
            }
        }
        if (!ATTR) {
#pragma unroll
            for (int v = 0; v < 3; v++)
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const float x = okv[v] ? a[v][ax] : __builtin_nanf("");
                    tri_s[s].p[3 * v + ax] = x;
                    tri_t[t].p[3 * v + ax] = x;
                }
        } else {
        float4 *row = reinterpret_cast<float4 *>(attr) + size_t(t) * 6;
        bool is_flat = all_ok;
#pragma unroll
        for (int v = 0; v < 3; v++) {
            row[2 * v] = make_float4(a[v][3], a[v][4], a[v][5], a[v][6]);
            row[2 * v + 1] = make_float4(a[v][7], a[v][8], 0.0f, 0.0f);
#pragma unroll
            for (int q = 3; q < 9; q++) is_flat = is_flat && (__float_as_uint(a[v][q]) == __float_as_uint(a[0][q]));
        }
        TriStatic r;
#pragma unroll
        for (int v = 0; v < 3; v++)
#pragma unroll
            for (int ax = 0; ax < 3; ax++) r.p[3 * v + ax] = okv[v] ? a[v][ax] : __builtin_nanf("");
#pragma unroll
        for (int q = 0; q < 3; q++) { r.n[q] = a[0][3 + q]; r.c[q] = a[0][6 + q]; }
        r.flags = (is_flat ? kFlagFlat : 0u) | (t << kFlagBits);
        tri_s[s] = r;
        tri_t[t] = r;
        }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++)
        for (int d = 32; d >= 1; d >>= 1) {
            lo[ax] = fminf(lo[ax], __shfl_xor(lo[ax], d));
            hi[ax] = fmaxf(hi[ax], __shfl_xor(hi[ax], d));
        }
    if ((threadIdx.x & 63u) == 0u)
        for (int ax = 0; ax < 3; ax++) { s_box[threadIdx.x >> 6][ax] = lo[ax]; s_box[threadIdx.x >> 6][3 + ax] = hi[ax]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = int(threadIdx.x);
        float v = s_box[0][k];
        for (int w = 1; w < 4; w++) v = k < 3 ? fminf(v, s_box[w][k]) : fmaxf(v, s_box[w][k]);
        box[size_t(blockIdx.x) * 6 + k] = v;
    }
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

// In-place vertex updates (ar_update_positions).  k_put_positions: new positions (three floats per vertex) into the
// position slots of the nine-float vertices.  k_flat_normals: every triangle's three vertices get the face normal exactly
// as k_build_vbo computes it (meant for meshes whose triangles do not share vertices).
__global__ __launch_bounds__(256) void k_put_positions(const float *__restrict__ positions, uint32_t n_verts, float *__restrict__ vbo) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n_verts) return;
    float *o = vbo + size_t(v) * 9;
    o[0] = positions[3 * size_t(v)]; o[1] = positions[3 * size_t(v) + 1]; o[2] = positions[3 * size_t(v) + 2];
}

__global__ __launch_bounds__(256) void k_flat_normals(float *__restrict__ vbo, const uint32_t *__restrict__ idx, uint32_t n_verts,
                                                      uint32_t n_tris) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tris) return;
    uint32_t vi[3];
    float p[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        vi[k] = idx[3 * size_t(t) + k];
        if (vi[k] >= n_verts) return;  // (such a triangle is invisible anyway)
#pragma unroll
        for (int ax = 0; ax < 3; ax++) p[k][ax] = vbo[size_t(vi[k]) * 9 + ax];
    }
    const double ux = double(p[1][0]) - double(p[0][0]), uy = double(p[1][1]) - double(p[0][1]), uz = double(p[1][2]) - double(p[0][2]);
    const double vx = double(p[2][0]) - double(p[0][0]), vy = double(p[2][1]) - double(p[0][1]), vz = double(p[2][2]) - double(p[0][2]);
    double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const double L = sqrt((nx * nx + ny * ny) + nz * nz);
    if (L > 1e-8) { nx /= L; ny /= L; nz /= L; } else { nx = 0.0; ny = 1.0; nz = 0.0; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float *o = vbo + size_t(vi[k]) * 9;
        o[3] = float(nx); o[4] = float(ny); o[5] = float(nz);
    }
}

// Copies the "sample the atlas" bit of every triangle's texel-coordinate row into the flags word of both of its
// records (cleared when triuv is null).  One lane per slot.
__global__ __launch_bounds__(256) void k_uvflags(const uint4 *__restrict__ triuv, const uint32_t *__restrict__ perm, uint32_t n_tris,
                                                 TriStatic *__restrict__ tri_s, TriStatic *__restrict__ tri_t) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= n_tris) return;
    const uint32_t t = perm[s];
    const bool sample = triuv && (triuv[t].w & kUvSample) != 0u;
    const uint32_t f = (tri_s[s].flags & ~kFlagTextured) | (sample ? kFlagTextured : 0u);
    tri_s[s].flags = f;
    tri_t[t].flags = f;
}

template <typename FP>
__device__ __forceinline__ Vert<0> transform_position(const FP &fp, float x, float y, float z) {
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
    by0 = (fp.rows - 1 - j1 - fp.ya) >> (fp.tshift + fp.bshift); by1 = (fp.rows - 1 - j0 - fp.ya) >> (fp.tshift + fp.bshift);
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
        if (!to_screen(t0.c, t1.c, t2.c, fp.hw, fp.hh, c[s].sc)) continue;
        const double det = facing_det(c[s].sc);
        if (!(det > 0.0)) continue;
        have[s] = cell_range(c[s].sc.X, c[s].sc.Y, c[s].sc.W, fp.cols, jlo, jhi, c[s].i0, c[s].i1, c[s].j0, c[s].j1);
    }
}

// Low word of a visibility key on the tiled path: the id, and below it whether the near plane cut the triangle
// (the order of ids is kept; the shading step then knows without the clip-space z of the vertices).
__device__ __forceinline__ uint32_t tile_key_low(uint32_t id, bool clipped) { return (id << 1) | uint32_t(clipped); }

// to_screen for a sub-triangle k_setup has binned: its vertices are known to be finite with w > 0.
__device__ __forceinline__ void to_screen_valid(const float (&c0)[4], const float (&c1)[4], const float (&c2)[4], float hw,
                                                float hh, Screen &s) {
    s.W[0] = c0[3]; s.W[1] = c1[3]; s.W[2] = c2[3];
    s.Z[0] = c0[2]; s.Z[1] = c1[2]; s.Z[2] = c2[2];
    s.X[0] = (c0[0] + c0[3]) * hw; s.X[1] = (c1[0] + c1[3]) * hw; s.X[2] = (c2[0] + c2[3]) * hw;
    s.Y[0] = (c0[1] + c0[3]) * hh; s.Y[1] = (c1[1] + c1[3]) * hh; s.Y[2] = (c2[1] + c2[3]) * hh;
}

// sub_triangle_screen (below) for a sub-triangle that k_setup has binned: no validity tests.
__device__ __forceinline__ void binned_triangle_screen(const FrameParams &fp, const float (&p)[9], uint32_t sub, Screen &sc,
                                                       bool &clipped) {
    const Vert<0> in0 = transform_position(fp, p[0], p[1], p[2]), in1 = transform_position(fp, p[3], p[4], p[5]),
                  in2 = transform_position(fp, p[6], p[7], p[8]);
    clipped = !(in0.c[2] + in0.c[3] >= 0.0f) || !(in1.c[2] + in1.c[3] >= 0.0f) || !(in2.c[2] + in2.c[3] >= 0.0f);
    if (!clipped) { to_screen_valid(in0.c, in1.c, in2.c, fp.hw, fp.hh, sc); return; }
    Vert<0> s0[3], s1[3];
    clip_near<0>(in0, in1, in2, s0, s1);
    if (sub) to_screen_valid(s1[0].c, s1[1].c, s1[2].c, fp.hw, fp.hh, sc);
    else to_screen_valid(s0[0].c, s0[1].c, s0[2].c, fp.hw, fp.hh, sc);
}

// Screen-space vertices of sub-triangle `sub` of the triangle with positions p: the transform, near clip and
// screen mapping of triangle_setup, for one sub-triangle.  clipped: the near plane cut the triangle.  False when
// the sub-triangle does not exist or cannot produce fragments (never the case for one that k_setup binned).
__device__ __forceinline__ bool sub_triangle_screen(const FrameParams &fp, const float (&p)[9], uint32_t sub, Screen &sc,
                                                    bool &clipped) {
    const Vert<0> in0 = transform_position(fp, p[0], p[1], p[2]), in1 = transform_position(fp, p[3], p[4], p[5]),
                  in2 = transform_position(fp, p[6], p[7], p[8]);
    clipped = !(in0.c[2] + in0.c[3] >= 0.0f) || !(in1.c[2] + in1.c[3] >= 0.0f) || !(in2.c[2] + in2.c[3] >= 0.0f);
    if (!clipped) return sub == 0u && to_screen(in0.c, in1.c, in2.c, fp.hw, fp.hh, sc);
    Vert<0> s0[3], s1[3];
    const int ns = clip_near<0>(in0, in1, in2, s0, s1);
    if (int(sub) >= ns) return false;
    return sub ? to_screen(s1[0].c, s1[1].c, s1[2].c, fp.hw, fp.hh, sc) : to_screen(s0[0].c, s0[1].c, s0[2].c, fp.hw, fp.hh, sc);
}

// Workgroup g owns survivor places [g * 2 * chunk, (g + 1) * 2 * chunk) and fills them densely, in the order it
// meets the survivors.  surv: the cell range of a survivor, (i0 | i1 << 16, j0 | j1 << 16); sid: 2*slot + sub -
// all k_fill needs.  Nothing else of the setup is kept: the tiles redo it from the triangle records.
// One triangle per lane and pass; chunk (a multiple of the workgroup size) bounds the passes of a workgroup.
// BATCH: one launch for several frames (parameters from fps, buffers at the frame's stride).
template <bool BATCH>
__global__ __launch_bounds__(kSetupThreads) void k_setup(FrameParams fp_one, const float *__restrict__ tripos, size_t stride,
                                                         const float *__restrict__ box, uint2 *__restrict__ surv,
                                                         uint32_t *__restrict__ sid, uint32_t *__restrict__ surv_count,
                                                         uint32_t *__restrict__ dircount, uint32_t *__restrict__ blocksum,
                                                         const FrameParams *__restrict__ fps, BatchStrides bs) {
    // Which workgroup of which frame this is.  One frame: the workgroup's own number.  Several frames in a launch: the
    // frames of one workgroup number are neighbours in the launch order and land on the same XCD (workgroups go to the
    // XCDs round-robin: w = ((g / 8) * frames + f) * 8 + g % 8), so that the block of positions frame 0 pulls into that
    // XCD's L2 serves the other frames' workgroups as well - every frame reads the same 36 bytes per triangle.
    uint32_t grp = blockIdx.x, frm = 0;
    if (BATCH) {
        const uint32_t q = blockIdx.x >> 3;
        frm = q % bs.frames;
        grp = (q / bs.frames) * 8u + (blockIdx.x & 7u);
        if (grp >= fps[frm].n_groups) return;
    }
    const FrameParams &fp = BATCH ? fps[frm] : fp_one;
    if (BATCH) {
        const size_t f = frm;
        surv += f * bs.surv; sid += f * bs.surv; surv_count += f * bs.groups; dircount += f * bs.dircount; blocksum += f * bs.dircount;
    }
    // (LDS sized by the launch to the frame's bins, not to the most a frame can have: the frame's kernels run beside
    // the tile resolve of the frame before, which leaves them a fifth of a compute unit's LDS)
    extern __shared__ uint32_t bins_lds[];
    uint32_t *hist = bins_lds;  // [nbins]
    __shared__ uint32_t s_wave[kSetupThreads / 64];
    __shared__ uint32_t outside[kMaxPasses];  // per pass: the frustum planes that all eight corners of its box lie beyond
    const uint32_t tid = threadIdx.x;
    const int nbins = fp.nbx * fp.nby;
    // Blocks of 256 slots are dealt to the workgroups round-robin (block p goes to workgroup p mod n_groups): the
    // slots are in spatial order, so a view or a strip that shows only part of the mesh leaves every workgroup
    // with about the same share of blocks that are not skipped below.
    const uint32_t total_pass = (fp.n_tris + kSetupThreads - 1) / kSetupThreads;
    const uint32_t n_pass = grp < total_pass ? (total_pass - grp + fp.n_groups - 1) / fp.n_groups : 0u;
    for (int k = int(tid); k < nbins; k += kSetupThreads) hist[k] = 0;
    for (uint32_t k = tid; k < n_pass; k += kSetupThreads) outside[k] = 0x3Fu;
    __syncthreads();
    const int jlo = fp.rows - fp.yb, jhi = fp.rows - 1 - fp.ya;
    // Which passes can be skipped: a pass covers one block of 256 slots, whose bounding box was taken at upload.
    // One lane per (pass, corner) puts the corner through the vertex transform and tests it against the six planes
    // that bound what this frame can show (columns -1 .. cols + 1, the strip's window rows widened by two, near,
    // far); a plane that every corner lies beyond - by more than the rounding of these sums can account for - has
    // the whole box, and every triangle in it, on its far side.
    for (uint32_t q = tid; q < 8u * n_pass; q += kSetupThreads) {
        const uint32_t pass = q >> 3;
        const float *b = box + (size_t(pass) * fp.n_groups + grp) * 6;
        const float x = b[(q & 1u) ? 3 : 0], y = b[(q & 2u) ? 4 : 1], z = b[(q & 4u) ? 5 : 2];
        float c[4], m[4];  // clip coordinates and the sums of the magnitudes of their terms
#pragma unroll
        for (int r = 0; r < 4; r++) {
            c[r] = ((fp.pv[0 * 4 + r] * x + fp.pv[1 * 4 + r] * y) + fp.pv[2 * 4 + r] * z) + fp.pv[3 * 4 + r];
            m[r] = ((fabsf(fp.pv[0 * 4 + r] * x) + fabsf(fp.pv[1 * 4 + r] * y)) + fabsf(fp.pv[2 * 4 + r] * z)) + fabsf(fp.pv[3 * 4 + r]);
        }
        constexpr float kEps = 1.0f / 1048576.0f;  // 2^-20: eight times what four rounded terms can be off by
        const float X = (c[0] + c[3]) * fp.hw, Y = (c[1] + c[3]) * fp.hh;
        const float mx = (m[0] + m[3]) * fp.hw, my = (m[1] + m[3]) * fp.hh;
        const float fj0 = float(jlo - 2), fj1 = float(jhi + 3), fc = float(fp.cols + 1);
        uint32_t o = 0;
        o |= (X + c[3] < -kEps * (mx + m[3])) ? 1u : 0u;                              // left of column -1
        o |= (fc * c[3] - X < -kEps * (mx + fc * m[3])) ? 2u : 0u;                     // right of column cols + 1
        o |= (Y - fj0 * c[3] < -kEps * (my + fabsf(fj0) * m[3])) ? 4u : 0u;            // below the strip's rows
        o |= (fj1 * c[3] - Y < -kEps * (my + fabsf(fj1) * m[3])) ? 8u : 0u;            // above them
        o |= (c[2] + c[3] < -kEps * (m[2] + m[3])) ? 16u : 0u;                         // behind the near plane
        o |= (c[3] - c[2] < -kEps * (m[2] + m[3])) ? 32u : 0u;                         // beyond the far plane
        atomicAnd(&outside[pass], o);
    }
    __syncthreads();
    const uint32_t loc_base = grp * 2u * fp.chunk;
    uint32_t run = 0;  // slots handed out so far (identical in every lane)

    for (uint32_t pass = 0; pass < n_pass; pass++) {
        if (outside[pass] != 0u) continue;  // (the same in every lane of the workgroup)
        const uint32_t t = (pass * fp.n_groups + grp) * kSetupThreads + tid;  // the slot: position in the spatial order
        const bool live = t < fp.n_tris;
        Candidate c[2];
        bool have[2] = {false, false}, clipped = false;
        if (live) {
            float p[9];
#pragma unroll
            for (int q = 0; q < 9; q++) p[q] = tripos[size_t(q) * stride + t];
            triangle_setup(fp, p, jlo, jhi, c, have, clipped);
        }
        uint32_t total;
        uint32_t slot = run + block_exclusive_scan_flags<kSetupThreads / 64>(have[0], have[1], s_wave, total);
        run += total;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (!have[s]) continue;
            const uint32_t loc = loc_base + slot;
            slot++;
            sid[loc] = 2u * t + uint32_t(s);
            surv[loc] = make_uint2(uint32_t(c[s].i0) | (uint32_t(c[s].i1) << 16), uint32_t(c[s].j0) | (uint32_t(c[s].j1) << 16));
            int bx0, bx1, by0, by1;
            bin_range(fp, c[s].i0, c[s].i1, c[s].j0, c[s].j1, bx0, bx1, by0, by1);
            for (int by = by0; by <= by1; by++)
                for (int bx = bx0; bx <= bx1; bx++) atomicAdd(&hist[by * fp.nbx + bx], 1u);
        }
    }
    __syncthreads();
    // The workgroup's row of the count table.  In a frame without k_binscan (fp.fold_scan) also its share of the totals
    // per bin and row block (16 row blocks of consecutive workgroups; blocksum is zero when such a frame starts): with
    // the triangles in spatial order a workgroup touches a handful of bins, so these are few atomics.
    const uint32_t my_block = grp / ((fp.n_groups + 15u) / 16u);
    for (int k = int(tid); k < nbins; k += kSetupThreads) {
        const uint32_t h = hist[k];
        dircount[size_t(grp) * nbins + k] = h;
        if (fp.fold_scan && h != 0u) atomicAdd(&blocksum[size_t(my_block) * nbins + k], h);
    }
    if (tid == 0) surv_count[grp] = run;
}

// Column-wise exclusive prefix of the count table over the k_setup workgroups, in two levels.  The
// table's rows are cut into 16 row blocks; workgroup (x, y) takes bins [64x, 64x + 64) of row block y
// (its four wavefronts split the block's rows), rewrites table[g][b] as "entries of bin b that belong to
// earlier workgroups of the same row block" and stores the block's total in blocksum[y][b].  k_fill
// adds the totals of the earlier row blocks.  No global atomics, and the layout of every bin's entry
// array is the same on every run.
// Four wavefronts per workgroup, not sixteen: the frame's kernels share the GPU with the tile resolve of the frame
// before, which leaves room for one or two wavefronts per SIMD - a workgroup that needs four of them on one
// compute unit waits until tiles there have finished (measured: 20 us, for every third frame).
constexpr int kScanWaves = 4;
constexpr int kFoldScanBins = 160;  // frames with at most so many bins go without k_binscan (FrameParams::fold_scan)
static_assert(kMaxGroups <= 16u * 16u * kScanWaves, "a wavefront of k_binscan takes at most 16 rows of its row block");
template <bool BATCH>
__global__ __launch_bounds__(64 * kScanWaves) void k_binscan(FrameParams fp_one, uint32_t *__restrict__ table, uint32_t *__restrict__ blocksum,
                                                              const FrameParams *__restrict__ fps, BatchStrides bs) {
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) { table += size_t(blockIdx.z) * bs.dircount; blocksum += size_t(blockIdx.z) * bs.dircount; }
    __shared__ uint32_t part[kScanWaves][64];
    const int nbins = fp.nbx * fp.nby;
    const int lane = int(threadIdx.x & 63u), w = int(threadIdx.x >> 6);
    const int b = int(blockIdx.x) * 64 + lane;
    const uint32_t rows_per_block = (fp.n_groups + 15u) / 16u;
    const uint32_t blk0 = min(uint32_t(blockIdx.y) * rows_per_block, fp.n_groups), blk1 = min(blk0 + rows_per_block, fp.n_groups);
    const uint32_t rows_per_wave = (rows_per_block + uint32_t(kScanWaves) - 1u) / uint32_t(kScanWaves);  // <= 16 (at most 1024 k_setup workgroups)
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
    for (int k = 0; k < kScanWaves; k++) {
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

// Where every bin's entry array starts, once per frame instead of once per k_fill workgroup (launches over several
// frames, FrameParams::bin_starts: there the extra launch is shared by all frames and what matters is the work saved -
// every one of k_fill's up to a thousand workgroups otherwise reads 16 x bins totals and scans them for itself).  One
// workgroup per frame: the 16 row-block totals per bin become, in place, "entries of earlier row blocks", the bins' totals
// are scanned over the bins, and start / total / overflow are published for k_fill and k_tile.
template <bool BATCH>
__global__ __launch_bounds__(kSetupThreads) void k_binstart(FrameParams fp_one, uint32_t *__restrict__ blocksum, BinState *__restrict__ state,
                                                            uint32_t entries_cap, const FrameParams *__restrict__ fps, BatchStrides bs) {
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) blocksum += size_t(blockIdx.z) * bs.dircount;
    __shared__ uint32_t cnt[kMaxBins];
    __shared__ uint32_t s_wave[kSetupThreads / 64];
    const uint32_t tid = threadIdx.x;
    const int nbins = fp.nbx * fp.nby;
    BinState &st = state[fp.set];
    for (int b = int(tid); b < nbins; b += kSetupThreads) {
        uint32_t x[16], total = 0;
#pragma unroll
        for (uint32_t y = 0; y < 16; y++) x[y] = blocksum[size_t(y) * nbins + b];
#pragma unroll
        for (uint32_t y = 0; y < 16; y++) {
            blocksum[size_t(y) * nbins + b] = total;
            total += x[y];
        }
        cnt[b] = total;
    }
    __syncthreads();
    const int kRun = (nbins + kSetupThreads - 1) / kSetupThreads;
    uint32_t run_total = 0;
    for (int q = 0; q < kRun; q++) {
        const int b = int(tid) * kRun + q;
        run_total += b < nbins ? cnt[b] : 0u;
    }
    uint32_t n_entries;
    uint32_t start = block_exclusive_scan<kSetupThreads / 64>(run_total, s_wave, n_entries);
    bool overflow = false;
    for (int q = 0; q < kRun; q++) {
        const int b = int(tid) * kRun + q;
        if (b >= nbins) break;
        const uint32_t total = cnt[b];
        const bool ok = start <= entries_cap && total <= entries_cap - start;  // the test k_fill and k_tile repeat
        overflow = overflow || (!ok && total != 0u);
        st.start[b] = start; st.total[b] = total;
        start += total;
    }
    const int any_overflow = __syncthreads_or(overflow);
    if (tid == 0) { st.n_entries = n_entries; st.overflow = any_overflow ? 1u : 0u; }
}

// Every workgroup first works out, from the 16 row-block totals per bin (k_binscan's blocksum), where each bin's
// entry array starts - total per bin, exclusive scan over the bins - and where its own entries go inside it:
// start of the bin + entries of earlier row blocks + entries of earlier workgroups of its own row block (the
// scanned table).  The totals are 16 x bins words read from L2 per workgroup; recomputing them here saves the frame
// a kernel whose only job was this scan.  Workgroup 0 also publishes start / total / overflow for k_tile.
template <bool BATCH>
__global__ __launch_bounds__(kSetupThreads) void k_fill(FrameParams fp_one, const TriStatic *__restrict__ tri,
                                                        const uint2 *__restrict__ surv, const uint32_t *__restrict__ sid,
                                                        const uint32_t *__restrict__ surv_count,
                                                        const uint32_t *__restrict__ dircount,
                                                        const uint32_t *__restrict__ blocksum, BinState *__restrict__ state,
                                                        uint32_t *__restrict__ entries, uint32_t entries_cap,
                                                        uint64_t *__restrict__ vis, const FrameParams *__restrict__ fps, BatchStrides bs) {
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) {
        const size_t f = blockIdx.z;
        surv += f * bs.surv; sid += f * bs.surv; surv_count += f * bs.groups; dircount += f * bs.dircount; blocksum += f * bs.dircount;
        entries += f * bs.entries; vis += f * bs.cells;
    }
    extern __shared__ uint32_t bins_lds[];  // (sized by the launch to the frame's bins, as in k_setup)
    const uint32_t tid = threadIdx.x;
    const int nbins = fp.nbx * fp.nby;
    uint32_t *base = bins_lds, *cnt = bins_lds + nbins;  // cnt holds the bin totals until the offsets are done
    uint8_t *fits = reinterpret_cast<uint8_t *>(bins_lds + 2 * nbins);
    __shared__ uint32_t s_wave[kSetupThreads / 64];
    BinState &st = state[fp.set];
    const uint32_t rows_per_block = (fp.n_groups + 15u) / 16u;
    const uint32_t my_block = blockIdx.x / rows_per_block;
    const uint32_t n = surv_count[blockIdx.x];
    if (fp.bin_starts) {
        // k_binstart has published the bins' starts and turned the row-block totals into prefixes
        if (n == 0u) return;  // (nothing to store, and nothing to publish)
        for (int b = int(tid); b < nbins; b += kSetupThreads) {
            const uint32_t start = st.start[b], total = st.total[b];
            base[b] = start + blocksum[size_t(my_block) * nbins + b] + dircount[size_t(blockIdx.x) * nbins + b];
            fits[b] = (start <= entries_cap && total <= entries_cap - start) ? 1 : 0;
            cnt[b] = 0;
        }
        __syncthreads();
    } else {
        for (int b = int(tid); b < nbins; b += kSetupThreads) {
            uint32_t total = 0, before = 0;
#pragma unroll
            for (uint32_t y = 0; y < 16; y++) {
                const uint32_t x = blocksum[size_t(y) * nbins + b];
                total += x;
                before += y < my_block ? x : 0u;
            }
            if (fp.fold_scan) {
                // no k_binscan has run: the table holds plain counts, and the entries of the earlier workgroups of this
                // row block are the sum of their rows (up to 63 rows of bins words from L2, sixteen loads in flight)
                for (uint32_t g0 = my_block * rows_per_block; g0 < blockIdx.x; g0 += 16u) {
                    uint32_t x[16];
#pragma unroll
                    for (uint32_t k = 0; k < 16; k++) x[k] = g0 + k < blockIdx.x ? dircount[size_t(g0 + k) * nbins + b] : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 16; k++) before += x[k];
                }
            } else {
                before += dircount[size_t(blockIdx.x) * nbins + b];  // (scanned by k_binscan)
            }
            cnt[b] = total;
            base[b] = before;
        }
        __syncthreads();
        // exclusive scan of the totals over the bins: each lane takes a run of consecutive bins (one bin per lane up
        // to 256 bins, which covers 1080p)
        const int kRun = (nbins + kSetupThreads - 1) / kSetupThreads;  // <= kMaxBins / kSetupThreads
        uint32_t run_total = 0;
        for (int q = 0; q < kRun; q++) {
            const int b = int(tid) * kRun + q;
            run_total += b < nbins ? cnt[b] : 0u;
        }
        uint32_t n_entries;
        uint32_t start = block_exclusive_scan<kSetupThreads / 64>(run_total, s_wave, n_entries);
        bool overflow = false;
        for (int q = 0; q < kRun; q++) {
            const int b = int(tid) * kRun + q;
            if (b >= nbins) break;
            const uint32_t total = cnt[b];
            const bool ok = start <= entries_cap && total <= entries_cap - start;
            base[b] += start;
            fits[b] = ok ? 1 : 0;
            overflow = overflow || (!ok && total != 0u);
            if (blockIdx.x == 0) { st.start[b] = start; st.total[b] = total; }
            start += total;
        }
        const int any_overflow = __syncthreads_or(overflow);
        if (blockIdx.x == 0 && tid == 0) { st.n_entries = n_entries; st.overflow = any_overflow ? 1u : 0u; }
        for (int b = int(tid); b < nbins; b += kSetupThreads) cnt[b] = 0;
        __syncthreads();
    }
    const uint32_t loc_base = blockIdx.x * 2u * fp.chunk;
    for (uint32_t s = tid; s < n; s += kSetupThreads) {
        const uint32_t loc = loc_base + s;
        const uint2 sv = surv[loc];
        const uint32_t id = sid[loc];  // 2*slot + sub: what the entries carry
        const int i0 = int(sv.x & 0xFFFFu), i1 = int(sv.x >> 16), j0 = int(sv.y & 0xFFFFu), j1 = int(sv.y >> 16);
        const int r0 = fp.rows - 1 - j1 - fp.ya, r1 = fp.rows - 1 - j0 - fp.ya;  // rows relative to the strip
        const int tx0 = i0 >> 6, tx1 = i1 >> 6, ty0 = r0 >> fp.tshift, ty1 = r1 >> fp.tshift;
        int bx0, bx1, by0, by1;
        bin_range(fp, i0, i1, j0, j1, bx0, bx1, by0, by1);
        // work class: rows of the cell range above its width, each clamped to its share of the class bits.  A lane of
        // k_tile walks a triangle's rows and then its candidate cells, and a wavefront takes as long as its slowest
        // lane: k_tile orders every tile's list by this class so that a wavefront's triangles are of one kind.  The
        // class steers the order of the work only, never a result.
        const int hb = min(fp.cls_bits, 3), wb = fp.cls_bits - hb;
        const uint32_t cls = uint32_t(((min(j1 - j0 + 1, 1 << hb) - 1) << wb) | (min(i1 - i0 + 1, 1 << wb) - 1)) << fp.id_bits;
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
                entries[size_t(base[b]) + atomicAdd(&cnt[b], 1u)] = id | cls | (m << kEntryMaskShift);
            }
        if (direct) {  // a bin it touches has no room: resolve through the global key buffer
            const TriStatic ts = tri[id >> 1];
            const uint32_t key_id = 2u * (ts.flags >> kFlagBits) + (id & 1u);  // keys carry the triangle's own index
            Screen sc;
            bool clipped;
            if (!sub_triangle_screen(fp, ts.p, id & 1u, sc, clipped)) continue;
            Edges e;
            edges_from_screen(sc, e);
            for (int j = j0; j <= j1; j++) {
                const size_t row = size_t(fp.rows - 1 - j) * fp.cols;
                for (int i = i0; i <= i1; i++) {
                    float f[3], zw;
                    if (cover(e, i, j, f, zw)) vis_min(vis, row + i, make_key(zw, tile_key_low(key_id, clipped)));
                }
            }
        }
    }
}

}  // namespace
