// ascii_stage.h - ASCII stage: quantiser, modal smoothing, text colours, glyph compositing
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
#pragma once

namespace {

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
// byte fields of registers.  Same Boyer-Moore order and vote rule as k_ascii (the plane holds no override cells,
// so every neighbour votes).
template <int R, bool BATCH>
__global__ __launch_bounds__(256) void k_ascii_plane(FrameParams fp_one, const uint8_t *__restrict__ src, uint8_t *__restrict__ ascii,
                                                     const FrameParams *__restrict__ fps, BatchStrides bs) {
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) { src += size_t(blockIdx.z) * bs.cells; ascii += size_t(blockIdx.z) * bs.ascii; }
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
        // Boyer-Moore with the counter kept as hits: every step adds +1 or -1 (a reset takes the new value as
        // candidate, which then matches itself), so after k neighbours the counter is 2 * hits - k and can only
        // be zero - the reset condition - when k is even and hits == k / 2.
        int cand = 0, hits = 0, votes = 0, k = 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                if (dx == 0 && dy == 0) continue;
                const int p = 4 + j + dx;
                const int ni = int((w[dy + R][p >> 2] >> (8 * (p & 3))) & 255u);
                if ((k & 1) == 0) cand = (k == 0 || hits == k / 2) ? ni : cand;
                hits += ni == cand ? 1 : 0;
                k++;
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

// Modal smoothing over a plane of ramp indices for short ramps (at most 10 characters) and radius 1 or 2: the same
// result as k_ascii_plane from counting instead of voting.  A lane owns four consecutive cells of a row.  It builds,
// for each of the 4 + 2R columns its windows touch, a histogram of the column's 2R + 1 values - ten 6-bit counters
// in one 64-bit word - so that a cell's window is a sum of 2R + 1 column words minus the cell itself, and the next
// cell's window is that sum plus one column minus another.  What the shader's two passes decide (the Boyer-Moore
// candidate, then its true vote count against the threshold) follows from the counts in all but one case:
//   no value reaches the threshold                       -> whatever the candidate is, it is not voted in: the cell stays;
//   some value holds a strict majority of the neighbours -> it is the candidate (that is what Boyer-Moore guarantees),
//                                                           and it has the votes whenever the threshold is at most
//                                                           its count - it replaces the cell;
//   values reach the threshold, none holds a majority    -> the candidate depends on the order of the neighbours: those
//                                                           cells are queued and a second step runs the shader's two
//                                                           passes on them as written, from the staged tile.
// The codes go through LDS so that the queued cells' bytes are in place before a lane stores its four.
// OV: the plane may hold kPlaneOverride (255) for cells whose alpha byte is a literal code (2..254: UI cells, atlas
// glyphs).  Such a cell is not smoothed - its code is the alpha byte of rgba - and does not vote: it adds nothing to
// the count words, and a sticky bit above the ten fields (set per column when any of its cells is one) sends every
// cell with one in its window to the second step, which runs the shader's passes with the cell skipped, as the shader
// does (ascii_pass_shader.js:98,125).  Overrides are sparse (a border, a line of text), so few cells take that road.
constexpr int kPlaneOverride = 255;
template <int R, bool OV, bool BATCH>
__global__ __launch_bounds__(256) void k_ascii_counts(FrameParams fp_one, const uint8_t *__restrict__ src, const uint32_t *__restrict__ rgba,
                                                      uint8_t *__restrict__ ascii, const FrameParams *__restrict__ fps, BatchStrides bs) {
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) { src += size_t(blockIdx.z) * bs.cells; ascii += size_t(blockIdx.z) * bs.ascii; }
    constexpr int TWD = 34, TH = 8 + 2 * R;  // as k_ascii_plane: dwords per tile row (4 bytes of halo either side), tile rows
    constexpr int N = (2 * R + 1) * (2 * R + 1) - 1, NC = 4 + 2 * R;
    constexpr unsigned long long kOnes = 0x0041041041041041ull;  // bit 0 of each of the ten 6-bit fields
    __shared__ uint32_t tile[TWD * TH];
    __shared__ uint8_t ramp[AR_MAX_RAMP];
    __shared__ uint32_t codes[256];   // the workgroup's 128 x 8 output codes, a lane's four in one word
    constexpr int NCOL = 128 + 2 * R;  // columns whose counts over a window's rows the block's windows need
    __shared__ unsigned long long colsum[8 * NCOL];  // [output row][column]: count word of the 2R + 1 cells of that column
    __shared__ uint16_t queue[1024];  // cells (row * 128 + column within the workgroup's block) left to the second step
    __shared__ uint32_t n_queue;
    const int x0 = int(blockIdx.x) * 128, y0 = fp.row0 + int(blockIdx.y) * 8;
    ramp[threadIdx.x] = fp.ramp[threadIdx.x];
    if (threadIdx.x == 0) n_queue = 0;
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
    const bool live = x < fp.cols && y < fp.row1;
    // field f of a count word holds how many of the neighbours have ramp index f.  Adding 32 - t to every field sets
    // its top bit exactly when the count is at least t (counts are at most N <= 24, so no field overflows).
    const int thresh = fp.mode_thresh;  // (the launcher sends only 1 <= thresh <= 32 here)
    const unsigned long long kTop = kOnes << 5;
    const unsigned long long k_thresh = kOnes * (unsigned long long)(32 - thresh);
    const unsigned long long k_major = kOnes * (unsigned long long)(32 - (N / 2 + 1));
    // First the counts per column: a window is 2R + 1 columns of 2R + 1 cells, and the column of a cell is shared by
    // the windows of 2R + 1 cells to its left and right and - one cell further down - by the rows below.  One work item
    // per (column, half of the block's rows): the count word of the first row's column, then one cell in and one out per
    // further row.  (Counted per window by the lane that owns it, the same words cost ten additions per cell; this way 2.4.)
    {
        const uint8_t *cells = reinterpret_cast<const uint8_t *>(tile);
        for (int item = int(threadIdx.x); item < 2 * NCOL; item += 256) {
            const int half = item >= NCOL ? 1 : 0, cb = item - half * NCOL;
            const uint8_t *column = cells + (4 * half) * (TWD * 4) + (4 - R) + cb;  // tile row 4 half: the first window row of output row 4 half
            // (a plane without override cells holds only ramp indices below ten: it is written by kernels of the same
            // frame under the same ramp - k_tile, k_shade, k_plane_from_rgba - and only rows written this frame are read)
            auto one = [](int v) { return OV ? (v < 10 ? 1ull << (6 * v) : 0ull) : 1ull << (6 * v); };
            unsigned long long acc = 0ull;
            int ovc = 0;  // override cells among the column's 2R + 1 (OV only)
#pragma unroll
            for (int t = 0; t <= 2 * R; t++) {
                const int v = int(column[t * (TWD * 4)]);
                acc += one(v);
                if (OV) ovc += v == kPlaneOverride ? 1 : 0;
            }
            colsum[(4 * half) * NCOL + cb] = acc | (OV && ovc ? 1ull << 60 : 0ull);
#pragma unroll
            for (int ro = 1; ro < 4; ro++) {
                const int vin = int(column[(ro + 2 * R) * (TWD * 4)]), vout = int(column[(ro - 1) * (TWD * 4)]);
                acc += one(vin);
                acc -= one(vout);
                if (OV) ovc += (vin == kPlaneOverride ? 1 : 0) - (vout == kPlaneOverride ? 1 : 0);
                colsum[(4 * half + ro) * NCOL + cb] = acc | (OV && ovc ? 1ull << 60 : 0ull);
            }
        }
    }
    __syncthreads();
    if (live) {
        unsigned long long col[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) col[c] = colsum[ly * NCOL + 4 * lx + c];  // columns x - R .. x + 3 + R
        const uint32_t centre = tile[(ly + R) * TWD + lx + 1];
        unsigned long long win = 0ull;
#pragma unroll
        for (int c = 0; c <= 2 * R; c++) win += col[c];
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int base = int((centre >> (8 * j)) & 255u);
            const unsigned long long here = win;
            if (j < 3) win += col[j + 2 * R + 1] - col[j];
            if (OV && base == kPlaneOverride) {  // a literal code: the alpha byte, never smoothed
                out |= (x + j < fp.cols ? rgba[size_t(y) * fp.cols + x + j] >> 24 : 0u) << (8 * j);
                continue;
            }
            if (OV && (here >> 60) != 0ull) {    // an override cell in the window: the shader's passes decide, without it
                queue[atomicAdd(&n_queue, 1u)] = uint16_t(ly * 128 + 4 * lx + j);
                continue;                        // (the second step writes this byte of the word in LDS)
            }
            const unsigned long long cnt = here - (1ull << (6 * base));  // the neighbours: the window without the cell itself
            int fin = base;
            const unsigned long long reach = (cnt + k_thresh) & kTop;
            if (reach != 0ull) {
                const unsigned long long major = (cnt + k_major) & kTop;
                const unsigned long long pick = thresh > N / 2 ? reach : major;
                if (pick != 0ull) {
                    fin = ((__ffsll((long long)pick) - 1) * 43) >> 8;  // top bit 6 f + 5 of the only such field -> f
                } else {
                    queue[atomicAdd(&n_queue, 1u)] = uint16_t(ly * 128 + 4 * lx + j);
                }
            }
            out |= uint32_t(fin < fp.ramp_len ? ramp[fin] : uint8_t(32)) << (8 * j);
        }
        codes[threadIdx.x] = out;
    }
    __syncthreads();
    // second step: the shader's two passes, as written, for the queued cells
    const uint32_t nq = n_queue;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(tile);
    for (uint32_t k = threadIdx.x; k < nq; k += 256) {
        const int cell = int(queue[k]), cy = cell >> 7, cx = cell & 127;
        int nb[N];
        {
            int q = 0;
#pragma unroll
            for (int dy = -R; dy <= R; dy++)
#pragma unroll
                for (int dx = -R; dx <= R; dx++)
                    if (dx != 0 || dy != 0) nb[q++] = bytes[(cy + R + dy) * (TWD * 4) + 4 + cx + dx];
        }
        const int base = bytes[(cy + R) * (TWD * 4) + 4 + cx];
        int cand = -1, count = 0, votes = 0;
#pragma unroll
        for (int q = 0; q < N; q++) {
            if (OV && nb[q] == kPlaneOverride) continue;  // override cells do not vote
            if (count == 0) { cand = nb[q]; count = 1; }
            else if (nb[q] == cand) count += 1;
            else count -= 1;
        }
#pragma unroll
        for (int q = 0; q < N; q++) votes += nb[q] == cand ? 1 : 0;
        const int fin = (cand >= 0 && votes >= thresh && cand != base) ? cand : base;
        reinterpret_cast<uint8_t *>(codes)[cell] = (fin >= 0 && fin < fp.ramp_len) ? ramp[fin] : uint8_t(32);
    }
    __syncthreads();
    if (!live) return;
    const uint32_t out = codes[threadIdx.x];
    uint8_t *dst = ascii + size_t(y) * fp.cols + x;
    if (cols4) {
        *reinterpret_cast<uint32_t *>(dst) = out;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x + j < fp.cols) dst[j] = uint8_t(out >> (8 * j));
    }
}

// Modal smoothing, bit-sliced: the counting kernel's decision for thresholds of at least half the neighbours (the
// reference's default: 12 of 24), taken for 56 cells at a time by one lane.  With such a threshold a cell changes only to
// a value that holds a strict majority of its neighbours (then Boyer-Moore's candidate is that value, whatever the order)
// and reaches the threshold; a value with exactly half of them - possible only when the threshold is exactly half - may or
// may not be the candidate, so those cells are redone by the shader's two passes as written; every other cell stays.
// Cells with a literal code (OV: plane value 255, index bits all set) belong to no ramp value, so they never vote; they
// keep their code, and a window with some of them has fewer voters, where "at least the threshold" is a majority all the more.
// Per ramp value k the cells holding k in one row are a bit mask; the count of k in a cell's window is a sum of shifted
// masks, done as bit-sliced arithmetic on 64-bit words (one bit position per cell; a full adder is two v_bitop3_b32 per
// 32 cells): per row the 2R + 1 horizontal neighbours (three bit planes), then a running sum over 2R + 1 rows (five
// planes: one row in, one out), minus the cell itself, compared against the threshold - about a hundred vector
// instructions per value, row and 56 cells, where the counting kernel spends eighty-five per cell.
// A wavefront takes a block of 168 cells x 2 RB rows: (1) a work item per (row, word) turns 64 bytes of ramp indices - the
// word's 56 cells and four either side for the windows' reach, grid-clamped like clampCell - into the four bit planes of
// the index; (2) a lane per (value, word, half of the rows) derives its value's mask from the planes row by row, keeps
// the sums in registers and ORs what it finds - "this value wins here", "exactly half here" - into per-row accumulators in
// LDS; (3) a work item per (row, word) merges winners into the planes, turns planes back into bytes, looks the glyphs up and
// stores 56 codes; (4) the cells with exactly half are queued and redone from the plane in memory.
// RB rows per lane: 16 where launches are large (many frames a launch: least work per cell), fewer for a single frame
// (more, shorter wavefronts).
constexpr int kBitsWords = 3, kBitsUse = 56;                       // words per block row, cells a word yields
constexpr unsigned long long kBitsUseful = 0x0FFFFFFFFFFFFFF0ull;  // bits 4 .. 59 of a word are its own cells
constexpr int kBitsQueue = 1024;
#ifndef AR_BITS_ROWS_SINGLE
#define AR_BITS_ROWS_SINGLE 8
#endif
constexpr int kBitsRowsSingle = AR_BITS_ROWS_SINGLE;  // rows per lane where one frame is smoothed per launch

// 64 cells as two registers; three-input boolean functions by truth table (inputs a = 0xF0, b = 0xCC, c = 0xAA)
struct Bits64 {
    uint32_t l, h;
};
template <int TT>
__device__ __forceinline__ Bits64 bits_op(Bits64 a, Bits64 b, Bits64 c) {
    return {uint32_t(__builtin_amdgcn_bitop3_b32(a.l, b.l, c.l, TT)), uint32_t(__builtin_amdgcn_bitop3_b32(a.h, b.h, c.h, TT))};
}
constexpr int kTtXor3 = 0x96, kTtMajority = 0xE8;
constexpr int kTtBorrow = 0x8E;   // borrow out of a - b - c
constexpr int kTtAndXor = 0x60;   // a & (b ^ c)
constexpr int kTtAtLeast = 0xD4;  // c ? a & b : a | b   (one bit of "number >= constant", from the low bit up)
constexpr int kTtSame = 0x90;     // a & ~(b ^ c)
constexpr int kTtSelect = 0xCA;   // a ? b : c
__device__ __forceinline__ Bits64 bits_xor(Bits64 a, Bits64 b) { return {a.l ^ b.l, a.h ^ b.h}; }
__device__ __forceinline__ Bits64 bits_and(Bits64 a, Bits64 b) { return {a.l & b.l, a.h & b.h}; }
__device__ __forceinline__ Bits64 bits_andn(Bits64 a, Bits64 b) {  // ~a & b
    return {uint32_t(__builtin_amdgcn_bitop3_b32(a.l, b.l, b.l, 0x0C)), uint32_t(__builtin_amdgcn_bitop3_b32(a.h, b.h, b.h, 0x0C))};
}
template <int TT>
__device__ __forceinline__ Bits64 bits_op1(Bits64 a, Bits64 b, uint32_t c) {  // the third input the same for both halves
    return {uint32_t(__builtin_amdgcn_bitop3_b32(a.l, b.l, c, TT)), uint32_t(__builtin_amdgcn_bitop3_b32(a.h, b.h, c, TT))};
}
__device__ __forceinline__ Bits64 bits_shr(Bits64 m, int d) { return {uint32_t(__builtin_amdgcn_alignbit(m.h, m.l, d)), m.h >> d}; }
__device__ __forceinline__ Bits64 bits_shl(Bits64 m, int d) { return {m.l << d, uint32_t(__builtin_amdgcn_alignbit(m.h, m.l, 32 - d))}; }
__device__ __forceinline__ Bits64 bits_load(const unsigned long long *p) {
    const unsigned long long v = *p;
    return {uint32_t(v), uint32_t(v >> 32)};
}

// The shader's two passes for one cell, as written, from the plane in memory (Boyer-Moore over the neighbours in window
// order, then the count of the candidate): the cell's ramp index after smoothing.
template <int R, bool OV>
__device__ __forceinline__ int ascii_exact_index(const FrameParams &fp, const uint8_t *__restrict__ src, int y, int xx) {
    constexpr int N = (2 * R + 1) * (2 * R + 1) - 1;
    int nbv[N];
    const uint8_t *centre = src + size_t(y) * fp.cols + xx;
    if (y >= R && y + R < fp.rows && xx >= R && xx + R < fp.cols) {  // no clamping away from the grid's edges
        int q = 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++)
                if (dx != 0 || dy != 0) nbv[q++] = centre[dy * fp.cols + dx];
    } else {
        int q = 0;
#pragma unroll
        for (int dy = -R; dy <= R; dy++)
#pragma unroll
            for (int dx = -R; dx <= R; dx++)
                if (dx != 0 || dy != 0)
                    nbv[q++] = src[size_t(min(max(y + dy, 0), fp.rows - 1)) * fp.cols + min(max(xx + dx, 0), fp.cols - 1)];
    }
    const int base = *centre;
    int cand = -1, count = 0, votes = 0;
#pragma unroll
    for (int q = 0; q < N; q++) {
        if (OV && nbv[q] == 255) continue;  // cells with a literal code do not vote
        if (count == 0) { cand = nbv[q]; count = 1; }
        else if (nbv[q] == cand) count += 1;
        else count -= 1;
    }
#pragma unroll
    for (int q = 0; q < N; q++) votes += nbv[q] == cand ? 1 : 0;
    return (votes >= fp.mode_thresh && cand != base) ? cand : base;
}

template <int R, int RB, bool OV, bool BATCH>
__global__ __launch_bounds__(64) void k_ascii_bits(FrameParams fp_one, const uint8_t *__restrict__ src, const uint32_t *__restrict__ rgba,
                                                   uint8_t *__restrict__ ascii, const FrameParams *__restrict__ fps, BatchStrides bs) {
    typedef unsigned long long u64;
    constexpr int ROWS = 2 * RB, TR = ROWS + 2 * R, N = (2 * R + 1) * (2 * R + 1) - 1, W = 2 * R + 1;
    __shared__ u64 planes[TR][kBitsWords][4];
    __shared__ u64 acc[ROWS][kBitsWords][6];  // any value wins, bits 0..3 of the winner, exactly half
    __shared__ uint16_t queue[kBitsQueue];     // cells with exactly half: block row << 8 | column within the block
    __shared__ uint32_t n_queue;
    const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;
    if (BATCH) { src += size_t(blockIdx.z) * bs.cells; ascii += size_t(blockIdx.z) * bs.ascii; }
    const int lane = int(threadIdx.x);
    const int x0 = int(blockIdx.x) * (kBitsWords * kBitsUse), y0 = fp.row0 + int(blockIdx.y) * ROWS;
    const bool cols4 = (fp.cols & 3) == 0;
    for (int k = lane; k < ROWS * kBitsWords * 6; k += 64) (&acc[0][0][0])[k] = 0ull;
    if (lane == 0) n_queue = 0;

    // (1) bytes -> index bit planes.  Bit i of word w is column x0 + 56 w - 4 + i of the row.
    for (int item = lane; item < TR * kBitsWords; item += 64) {
        const int tr = item / kBitsWords, w = item - tr * kBitsWords;
        const int gy = min(max(y0 + tr - R, 0), fp.rows - 1), cs = x0 + kBitsUse * w - 4;
        if (cs + 4 >= fp.cols) continue;  // none of the word's own cells is on the grid
        const uint8_t *row = src + size_t(gy) * fp.cols;
        uint32_t d[16];
        if (cols4 && cs >= 0 && cs + 64 <= fp.cols) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = reinterpret_cast<const uint4 *>(row + cs)[q];  // cs, cols: multiples of 4 -> 4-byte aligned
                d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v |= uint32_t(row[min(max(cs + 4 * q + b, 0), fp.cols - 1)]) << (8 * b);
                d[q] = v;
            }
        }
        // the low nibbles of eight cells to a register, cell c at bits 4c .. 4c + 3
        uint32_t nib[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint32_t a = d[2 * q], b = d[2 * q + 1];
            if (OV) { a &= 0x0F0F0F0Fu; b &= 0x0F0F0F0Fu; }
            a = (a | (a >> 4)) & 0x00FF00FFu; b = (b | (b >> 4)) & 0x00FF00FFu;
            a = (a | (a >> 8)) & 0x0000FFFFu; b = b | (b >> 8);
            nib[q] = a | (b << 16);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t half[2] = {0u, 0u};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                // bit j of each nibble, the eight of them gathered into a byte
                uint32_t t = (nib[q] >> j) & 0x11111111u;
                t = (t | (t >> 3)) & 0x03030303u;
                t = (t | (t >> 6)) & 0x000F000Fu;
                t = (t | (t >> 12)) & 0xFFu;
                half[q >> 2] |= t << (8 * (q & 3));
            }
            planes[tr][w][j] = u64(half[0]) | (u64(half[1]) << 32);
        }
    }
    __syncthreads();

    // (2) a lane per (ramp value, word, half of the block's rows)
    const int thresh = fp.mode_thresh;
    const int need = thresh > N / 2 + 1 ? thresh : N / 2 + 1;  // votes a value needs: a strict majority that reaches the threshold
    const bool halves = thresh <= N / 2;                       // exactly half can reach the threshold: the order of the neighbours decides
    if (lane < 10 * kBitsWords * 2) {
        const int k = lane / (kBitsWords * 2), rem = lane - k * (kBitsWords * 2);
        const int w = rem % kBitsWords, rg = rem / kBitsWords;
        if (k < fp.ramp_len && y0 + rg * RB < fp.row1 && x0 + kBitsUse * w < fp.cols) {
            // x[j]: all ones where bit j of k is clear, so that (plane ^ x) is "bit j agrees with k"
            uint32_t x[4], kbit[4], need_bit[5], half_bit[5];
#pragma unroll
            for (int j = 0; j < 4; j++) { kbit[j] = ((k >> j) & 1) ? ~0u : 0u; x[j] = ~kbit[j]; }
#pragma unroll
            for (int q = 0; q < 5; q++) {
                need_bit[q] = ((need >> q) & 1) ? ~0u : 0u;
                half_bit[q] = (((N / 2) >> q) & 1) ? ~0u : 0u;
            }
            const u64 *pl = &planes[rg * RB][w][0];
            u64 *out = &acc[rg * RB][w][0];
            const Bits64 useful = {uint32_t(kBitsUseful), uint32_t(kBitsUseful >> 32)};
            Bits64 h0[W], h1[W], h2[W];  // horizontal sums of the last 2R + 1 rows (ring, statically indexed: the loop is unrolled)
            Bits64 own[R + 1];           // masks of the last R + 1 rows: the centre row's is the cell itself
            Bits64 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0}, s4 = {0, 0};  // the window's sum
#pragma unroll
            for (int t = 0; t < RB + 2 * R; t++) {
                Bits64 m = bits_load(pl + t * kBitsWords * 4);
                m.l ^= x[0]; m.h ^= x[0];
#pragma unroll
                for (int j = 1; j < 4; j++) m = bits_op1<kTtAndXor>(m, bits_load(pl + t * kBitsWords * 4 + j), x[j]);
                own[t % (R + 1)] = m;
                Bits64 a0, a1, a2;
                if (R == 2) {
                    const Bits64 r2 = bits_shr(m, 2), r1 = bits_shr(m, 1), l1 = bits_shl(m, 1), l2 = bits_shl(m, 2);
                    const Bits64 p = bits_op<kTtXor3>(r2, r1, m), c1 = bits_op<kTtMajority>(r2, r1, m);
                    a0 = bits_op<kTtXor3>(p, l1, l2);
                    const Bits64 c2 = bits_op<kTtMajority>(p, l1, l2);
                    a1 = bits_xor(c1, c2); a2 = bits_and(c1, c2);
                } else {
                    const Bits64 r1 = bits_shr(m, 1), l1 = bits_shl(m, 1);
                    a0 = bits_op<kTtXor3>(r1, m, l1); a1 = bits_op<kTtMajority>(r1, m, l1);
                    a2 = {0, 0};
                }
                if (t >= W) {  // the row that leaves the window
                    const Bits64 o0 = h0[t % W], o1 = h1[t % W], o2 = h2[t % W];
                    Bits64 b = bits_andn(s0, o0), n;
                    s0 = bits_xor(s0, o0);
                    n = bits_op<kTtXor3>(s1, o1, b); b = bits_op<kTtBorrow>(s1, o1, b); s1 = n;
                    n = bits_op<kTtXor3>(s2, o2, b); b = bits_op<kTtBorrow>(s2, o2, b); s2 = n;
                    n = bits_xor(s3, b); b = bits_andn(s3, b); s3 = n;
                    s4 = bits_xor(s4, b);
                }
                h0[t % W] = a0; h1[t % W] = a1; h2[t % W] = a2;
                {   // the row that enters
                    Bits64 c = bits_and(s0, a0), n;
                    s0 = bits_xor(s0, a0);
                    n = bits_op<kTtXor3>(s1, a1, c); c = bits_op<kTtMajority>(s1, a1, c); s1 = n;
                    n = bits_op<kTtXor3>(s2, a2, c); c = bits_op<kTtMajority>(s2, a2, c); s2 = n;
                    n = bits_xor(s3, c); c = bits_and(s3, c); s3 = n;
                    s4 = bits_xor(s4, c);
                }
                if (t >= 2 * R) {  // the window of row t - 2R of the lane's rows is complete (rows below the strip: computed, never stored)
                    // neighbours = window - the cell itself
                    const Bits64 me = own[(t - R) % (R + 1)];
                    Bits64 nb[5], b = bits_andn(s0, me);
                    nb[0] = bits_xor(s0, me);
                    nb[1] = bits_xor(s1, b); b = bits_andn(s1, b);
                    nb[2] = bits_xor(s2, b); b = bits_andn(s2, b);
                    nb[3] = bits_xor(s3, b); b = bits_andn(s3, b);
                    nb[4] = bits_xor(s4, b);
                    Bits64 ge = useful, eq = useful;
#pragma unroll
                    for (int q = 0; q < 5; q++) ge = bits_op1<kTtAtLeast>(nb[q], ge, need_bit[q]);
                    if (need > N) ge = {0, 0};
                    ge = bits_and(ge, useful);
                    if (halves) {
#pragma unroll
                        for (int q = 0; q < 5; q++) eq = bits_op1<kTtSame>(eq, nb[q], half_bit[q]);
                        // a cell that holds the value itself stays whatever the candidate is - unless a second value has
                        // exactly half too, and then that value's lane reports the cell
                        eq = bits_andn(me, eq);
                    } else {
                        eq = {0, 0};
                    }
                    if ((ge.l | ge.h | eq.l | eq.h) != 0u) {
                        u64 *o = out + (t - 2 * R) * kBitsWords * 6;
                        atomicOr(o, u64(ge.l) | (u64(ge.h) << 32));
#pragma unroll
                        for (int j = 0; j < 4; j++) atomicOr(o + 1 + j, u64(ge.l & kbit[j]) | (u64(ge.h & kbit[j]) << 32));
                        atomicOr(o + 5, u64(eq.l) | (u64(eq.h) << 32));
                    }
                }
            }
        }
    }
    __syncthreads();

    // (3) the cells with exactly half: queued, so that all lanes share them; what the two passes decide is merged into
    // the winners (no value holds a majority there, so the cell's winner bits are still clear)
    auto resolve = [&](int ro, int col) {
        const int y = y0 + ro, w = col / kBitsUse, bit = col - w * kBitsUse + 4;
        const int fin = ascii_exact_index<R, OV>(fp, src, y, x0 + col);
        if (fin != int(src[size_t(y) * fp.cols + x0 + col])) {
            atomicOr(&acc[ro][w][0], 1ull << bit);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((fin >> j) & 1) atomicOr(&acc[ro][w][1 + j], 1ull << bit);
        }
    };
    if (halves) {
        for (int item = lane; item < ROWS * kBitsWords; item += 64) {
            const int ro = item / kBitsWords, w = item - ro * kBitsWords;
            if (y0 + ro >= fp.row1) continue;
            u64 todo = acc[ro][w][5];
            if (OV) todo &= ~(planes[ro + R][w][0] & planes[ro + R][w][1] & planes[ro + R][w][2] & planes[ro + R][w][3]);  // literal codes stay
            while (todo != 0ull) {
                const int i = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int col = kBitsUse * w + i - 4;  // column within the block
                if (x0 + col >= fp.cols) continue;
                const uint32_t place = atomicAdd(&n_queue, 1u);
                if (place < uint32_t(kBitsQueue)) queue[place] = uint16_t((ro << 8) | col);
                else resolve(ro, col);  // queue full (a checkerboard has every cell at exactly half): this lane does it
            }
        }
        __syncthreads();
        const uint32_t nq = min(n_queue, uint32_t(kBitsQueue));
        for (uint32_t e = lane; e < nq; e += 64) resolve(int(queue[e] >> 8), int(queue[e] & 255u));
        __syncthreads();
    }

    // (4) winners into the planes, planes into bytes, bytes into glyphs
    uint32_t ramp_lo = 0, ramp_mid = 0, ramp_hi = 0;  // glyphs 0..3, 4..7, 8..9 as bytes
#pragma unroll
    for (int q = 0; q < 4; q++) {
        ramp_lo |= uint32_t(fp.ramp[q]) << (8 * q);
        ramp_mid |= uint32_t(fp.ramp[4 + q]) << (8 * q);
        if (q < 2) ramp_hi |= uint32_t(fp.ramp[8 + q]) << (8 * q);
    }
    for (int item = lane; item < ROWS * kBitsWords; item += 64) {
        const int ro = item / kBitsWords, w = item - ro * kBitsWords;
        const int y = y0 + ro, cx = x0 + kBitsUse * w;  // first of the word's own columns
        if (y >= fp.row1 || cx >= fp.cols) continue;
        u64 any = acc[ro][w][0], lit = 0ull;
        if (OV) {  // cells with a literal code (index bits all set) keep it
            lit = planes[ro + R][w][0] & planes[ro + R][w][1] & planes[ro + R][w][2] & planes[ro + R][w][3] & kBitsUseful;
            any &= ~lit;
        }
        const Bits64 any2 = {uint32_t(any), uint32_t(any >> 32)};
        Bits64 f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = bits_op<kTtSelect>(any2, bits_load(&acc[ro][w][1 + j]), bits_load(&planes[ro + R][w][j]));
        uint8_t *dst = ascii + size_t(y) * fp.cols + cx;
        uint32_t code[kBitsUse / 4];
#pragma unroll
        for (int q = 0; q < kBitsUse / 4; q++) {
            const int pos = 4 + 4 * q;
            uint32_t idx4 = 0;  // four ramp indices as bytes
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t four = ((pos < 32 ? f[j].l >> pos : f[j].h >> (pos - 32)) & 15u);
                idx4 |= (__umul24(four, 0x00204081u) & 0x01010101u) << j;  // bit b of the four -> bit j of byte b
            }
            // glyph lookup: v_perm picks bytes 0..7 of (mid : lo) by the index; indices 8, 9 come from the third word
            const uint32_t low8 = __builtin_amdgcn_perm(ramp_mid, ramp_lo, idx4 & 0x07070707u);
            const uint32_t top = __builtin_amdgcn_perm(0u, ramp_hi, idx4 & 0x01010101u);
            const uint32_t top_bit = (idx4 >> 3) & 0x01010101u, is_top = (top_bit << 8) - top_bit;  // 255 in the bytes with index 8 or 9
            uint32_t code4 = (top & is_top) | (low8 & ~is_top);
            if (OV) {
                const uint32_t l4 = uint32_t(lit >> pos) & 15u;
                if (l4 != 0u) {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (((l4 >> b) & 1u) && cx + 4 * q + b < fp.cols)
                            code4 = (code4 & ~(255u << (8 * b))) | ((rgba[size_t(y) * fp.cols + cx + 4 * q + b] >> 24) << (8 * b));
                }
            }
            code[q] = code4;
        }
        if (cols4 && cx + kBitsUse <= fp.cols) {  // 56 bytes at a 4-byte boundary: three 16-byte stores and one of 8
#pragma unroll
            for (int q = 0; q < 3; q++) reinterpret_cast<uint4 *>(dst)[q] = make_uint4(code[4 * q], code[4 * q + 1], code[4 * q + 2], code[4 * q + 3]);
            reinterpret_cast<uint2 *>(dst)[6] = make_uint2(code[12], code[13]);
        } else {
#pragma unroll
            for (int q = 0; q < kBitsUse / 4; q++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (cx + 4 * q + b < fp.cols) dst[4 * q + b] = uint8_t(code[q] >> (8 * b));
        }
    }
}

// RGBA8 grid -> plane of ramp indices over rows [ya, yb), kPlaneOverride where the alpha byte is a literal code: the input
// of the counting kernel for grids that do not come with a plane (AsciiPass.drawFromBuffer's input, ray frames).
__global__ __launch_bounds__(256) void k_plane_from_rgba(FrameParams fp, const uint32_t *__restrict__ px, uint8_t *__restrict__ plane) {
    const size_t first = size_t(fp.ya) * fp.cols, n = size_t(fp.yb - fp.ya) * fp.cols;
    const size_t p = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t me = px[first + p];
    const int a = int(me >> 24);
    plane[first + p] = (a >= 2 && a <= 254) ? uint8_t(kPlaneOverride)
                                            : uint8_t(ramp_index(int(me & 255u), int((me >> 8) & 255u), int((me >> 16) & 255u), fp.ramp_len));
}

// The counting kernel's domain: ramps of at most ten glyphs, radius 1 or 2, thresholds that fit a field.
inline bool ascii_counts_domain(const FrameParams &fp) {
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 3 ? 3 : fp.mode_radius);
    return fp.ramp_len <= 10 && radius <= 2 && fp.mode_thresh >= 1 && fp.mode_thresh <= 32;
}

// Smoothing of an RGBA8 grid that may carry literal codes in its alpha bytes: through a plane of ramp indices with the
// override cells marked (plane: scratch of cols * rows bytes; already filled when plane_ready - k_tile writes one for
// textured frames) and the counting kernel when the parameters allow, else the generic kernel straight from the grid.
inline void launch_ascii_rgba(const FrameParams &fp, const uint32_t *rgba, uint8_t *plane, bool plane_ready, uint8_t *ascii, hipStream_t stream);

// Smoothing of n frames' ramp-index planes in one launch (n = 1: the single frame of fp; else fps holds the n frames'
// parameters and the planes lie bs.cells / bs.ascii bytes apart).
// The bit-sliced kernel decides by "strict majority that reaches the threshold", which is the shader's answer only when the
// threshold is at least half the neighbours.  Its wavefronts are long (a block of 168 x 32 cells each): in a launch over
// several frames, or behind the kernel that makes the plane of an RGBA grid, it is the faster one; for one rendered frame
// on an otherwise idle device the counting kernel finishes sooner (many short workgroups), so `single` launches take the
// bit-sliced kernel only when asked to (ascii_path 3).
inline bool ascii_bits_domain(const FrameParams &fp, bool single) {
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 3 ? 3 : fp.mode_radius);
    const int n = (2 * radius + 1) * (2 * radius + 1) - 1;
    return ascii_counts_domain(fp) && 2 * fp.mode_thresh >= n && (fp.ascii_path == 3 || (fp.ascii_path == 0 && !single));
}

inline void launch_ascii_planes(const FrameParams &fp, const uint8_t *plane, uint8_t *ascii, hipStream_t stream, int n,
                                const FrameParams *fps, const BatchStrides &bs) {
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 3 ? 3 : fp.mode_radius);  // shader clamps to 1..3
    const dim3 grid((fp.cols + 127) / 128, (fp.row1 - fp.row0 + 7) / 8, unsigned(n));
    const int gbx = (fp.cols + kBitsWords * kBitsUse - 1) / (kBitsWords * kBitsUse), strip = fp.row1 - fp.row0;
    const dim3 gbits1(gbx, (strip + 2 * kBitsRowsSingle - 1) / (2 * kBitsRowsSingle), 1);
    const bool counts = ascii_counts_domain(fp) && fp.ascii_path != 2;
    if (n > 1) {
        if (ascii_bits_domain(fp, false)) {
            // rows per lane: 16 do the least work per cell, but a launch of few, long wavefronts (strips, small grids)
            // leaves most of the device idle - fewer rows per lane until there are a few thousand wavefronts
            auto blocks = [&](int rb) { return size_t(gbx) * size_t((strip + 2 * rb - 1) / (2 * rb)) * size_t(n); };
            const int rb = blocks(16) >= 4096 ? 16 : (blocks(8) >= 4096 ? 8 : 4);
            const dim3 gbits(gbx, (strip + 2 * rb - 1) / (2 * rb), unsigned(n));
            if (radius == 1) {
                if (rb == 16) k_ascii_bits<1, 16, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
                else if (rb == 8) k_ascii_bits<1, 8, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
                else k_ascii_bits<1, 4, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
            } else {
                if (rb == 16) k_ascii_bits<2, 16, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
                else if (rb == 8) k_ascii_bits<2, 8, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
                else k_ascii_bits<2, 4, false, true><<<gbits, 64, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
            }
        } else if (counts) {
            if (radius == 1) k_ascii_counts<1, false, true><<<grid, 256, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
            else k_ascii_counts<2, false, true><<<grid, 256, 0, stream>>>(fp, plane, nullptr, ascii, fps, bs);
        } else if (radius == 1) k_ascii_plane<1, true><<<grid, 256, 0, stream>>>(fp, plane, ascii, fps, bs);
        else if (radius == 2) k_ascii_plane<2, true><<<grid, 256, 0, stream>>>(fp, plane, ascii, fps, bs);
        else k_ascii_plane<3, true><<<grid, 256, 0, stream>>>(fp, plane, ascii, fps, bs);
        return;
    }
    if (ascii_bits_domain(fp, true)) {
        if (radius == 1) k_ascii_bits<1, kBitsRowsSingle, false, false><<<gbits1, 64, 0, stream>>>(fp, plane, nullptr, ascii, nullptr, bs);
        else k_ascii_bits<2, kBitsRowsSingle, false, false><<<gbits1, 64, 0, stream>>>(fp, plane, nullptr, ascii, nullptr, bs);
        return;
    }
    if (counts) {
        if (radius == 1) k_ascii_counts<1, false, false><<<grid, 256, 0, stream>>>(fp, plane, nullptr, ascii, nullptr, bs);
        else k_ascii_counts<2, false, false><<<grid, 256, 0, stream>>>(fp, plane, nullptr, ascii, nullptr, bs);
        return;
    }
    if (radius == 1) k_ascii_plane<1, false><<<grid, 256, 0, stream>>>(fp, plane, ascii, nullptr, bs);
    else if (radius == 2) k_ascii_plane<2, false><<<grid, 256, 0, stream>>>(fp, plane, ascii, nullptr, bs);
    else k_ascii_plane<3, false><<<grid, 256, 0, stream>>>(fp, plane, ascii, nullptr, bs);
}

template <bool FROM_RGBA>
void launch_ascii(const FrameParams &fp, const void *src, uint8_t *ascii, hipStream_t stream) {
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 3 ? 3 : fp.mode_radius);  // shader clamps to 1..3
    if constexpr (!FROM_RGBA) {
        launch_ascii_planes(fp, static_cast<const uint8_t *>(src), ascii, stream, 1, nullptr, BatchStrides{});
    } else {
        const dim3 grid((fp.cols + 31) / 32, (fp.row1 - fp.row0 + 7) / 8);
        if (radius == 1) k_ascii<1, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
        else if (radius == 2) k_ascii<2, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
        else k_ascii<3, FROM_RGBA><<<grid, 256, 0, stream>>>(fp, src, ascii);
    }
}

inline void launch_ascii_rgba(const FrameParams &fp, const uint32_t *rgba, uint8_t *plane, bool plane_ready, uint8_t *ascii, hipStream_t stream) {
    if (!plane || !ascii_counts_domain(fp) || fp.ascii_path == 2) {
        launch_ascii<true>(fp, rgba, ascii, stream);
        return;
    }
    const int radius = fp.mode_radius < 1 ? 1 : (fp.mode_radius > 2 ? 2 : fp.mode_radius);
    if (!plane_ready) {
        const size_t n = size_t(fp.yb - fp.ya) * fp.cols;
        k_plane_from_rgba<<<unsigned((n + 255) / 256), 256, 0, stream>>>(fp, rgba, plane);
    }
    if (ascii_bits_domain(fp, false)) {
        const dim3 gbits((fp.cols + kBitsWords * kBitsUse - 1) / (kBitsWords * kBitsUse), (fp.row1 - fp.row0 + 2 * kBitsRowsSingle - 1) / (2 * kBitsRowsSingle));
        if (radius == 1) k_ascii_bits<1, kBitsRowsSingle, true, false><<<gbits, 64, 0, stream>>>(fp, plane, rgba, ascii, nullptr, BatchStrides{});
        else k_ascii_bits<2, kBitsRowsSingle, true, false><<<gbits, 64, 0, stream>>>(fp, plane, rgba, ascii, nullptr, BatchStrides{});
        return;
    }
    const dim3 grid((fp.cols + 127) / 128, (fp.row1 - fp.row0 + 7) / 8);
    if (radius == 1) k_ascii_counts<1, true, false><<<grid, 256, 0, stream>>>(fp, plane, rgba, ascii, nullptr, BatchStrides{});
    else k_ascii_counts<2, true, false><<<grid, 256, 0, stream>>>(fp, plane, rgba, ascii, nullptr, BatchStrides{});
}

}  // namespace
