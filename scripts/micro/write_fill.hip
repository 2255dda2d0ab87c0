// Microbenchmark: do plain stores make the L2 read the lines they write?  A plane of 1920 x 1080 cells x 4 bytes is written
// (a) linearly, 16 bytes per lane; (b) tile by tile as the tile resolve does - a workgroup of 256 lanes writes a 64 x 32-cell
// tile, 16 lanes per 256-byte row segment, rows 7680 bytes apart; (c) as (b) but the workgroup first spends a while on
// arithmetic between the rows (stores of one tile spread over time); (d) as (b) with a second, unrelated read stream
// running through the same L2s.  Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum and compare reads per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kCols = 1920, kRows = 1080;

__global__ void k_linear(uint4 *out, size_t n16) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n16) out[i] = make_uint4(uint32_t(i), 1, 2, 3);
}

template <int SPIN, bool READS>
__global__ void k_tiles(uint32_t *out, const uint4 *__restrict__ other, size_t other_n, uint32_t *sink) {
    const int tx = blockIdx.x % 30, ty = blockIdx.x / 30, frame = blockIdx.y;
    uint32_t *plane = out + size_t(frame) * kCols * kRows;
    uint32_t acc = threadIdx.x;
    for (int q = int(threadIdx.x); q < 64 * 32 / 4; q += 256) {
        const int row = q >> 4, x = (q & 15) * 4;
        const int rr = ty * 32 + row, i = tx * 64 + x;
        if (rr >= kRows) continue;
        if (READS) {
            const uint4 v = other[(size_t(blockIdx.x) * 977 + size_t(q) * 131 + size_t(frame) * 7919) % other_n];
            acc ^= v.x + v.w;
        }
        for (int s = 0; s < SPIN; s++) acc = acc * 1664525u + 1013904223u;
        reinterpret_cast<uint4 *>(plane)[(size_t(rr) * kCols + i) >> 2] = make_uint4(acc, 1, 2, 3);
    }
    if (acc == 0x12345u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int frames = 16;
    const size_t cells = size_t(kCols) * kRows, bytes = cells * 4 * frames;
    uint32_t *out, *sink; uint4 *other;
    const size_t other_n = size_t(1) << 24;  // 256 MiB of 16-byte pieces
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&other, other_n * 16));
    CK(hipMemset(other, 1, other_n * 16));
    for (int rep = 0; rep < 3; rep++) {
        k_linear<<<unsigned((bytes / 16 + 255) / 256), 256>>>(reinterpret_cast<uint4 *>(out), bytes / 16);
        k_tiles<0, false><<<dim3(30 * 34, frames), 256>>>(out, other, other_n, sink);
        k_tiles<2000, false><<<dim3(30 * 34, frames), 256>>>(out, other, other_n, sink);
        k_tiles<0, true><<<dim3(30 * 34, frames), 256>>>(out, other, other_n, sink);
    }
    CK(hipDeviceSynchronize());
    printf("done: %zu bytes per launch = %zu lines of 128 bytes\n", bytes, bytes / 128);
    return 0;
}
