// Microbenchmark: what does a scattered read of 64 bytes cost next to one of 128 bytes?
// N random, distinct 128-byte lines of a buffer larger than every cache; one lane per line reads
// (a) the first 64 bytes, (b) all 128 bytes, (c) 16 bytes.  If (a) and (b) take the same time the
// memory side moves whole 128-byte lines and a 64-byte record wastes half of every fetch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int NV>  // number of 16-byte pieces read per lane
__global__ void k_read(const uint4 *__restrict__ buf, const uint32_t *__restrict__ line, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint4 *p = buf + size_t(line[i]) * 8;
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < NV; q++) { const uint4 v = p[q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t lines = size_t(1) << 23;  // 1 GiB of 128-byte lines
    const uint32_t n = 1u << 20;           // lines touched per launch
    uint4 *buf; uint32_t *idx, *out;
    CK(hipMalloc(&buf, lines * 128)); CK(hipMemset(buf, 1, lines * 128));
    CK(hipMalloc(&idx, n * 4 * 8)); CK(hipMalloc(&out, 4));
    std::vector<uint32_t> all(lines); std::iota(all.begin(), all.end(), 0u);
    std::mt19937 rng(1); std::shuffle(all.begin(), all.end(), rng);
    CK(hipMemcpy(idx, all.data(), size_t(n) * 4 * 8, hipMemcpyHostToDevice));  // 8 disjoint sets of n distinct lines
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](int nv, int set) {
        const uint32_t *li = idx + size_t(set) * n;
        hipEventRecord(a);
        if (nv == 1) k_read<1><<<n / 256, 256>>>(buf, li, n, out);
        else if (nv == 4) k_read<4><<<n / 256, 256>>>(buf, li, n, out);
        else k_read<8><<<n / 256, 256>>>(buf, li, n, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); return ms;
    };
    run(1, 7);  // warm the code
    int set = 0;
    for (int rep = 0; rep < 2; rep++)
        for (int nv : {1, 4, 8}) {
            const float ms = run(nv, set++ % 7);
            printf("%3d bytes per line: %7.1f us for %u scattered lines, %.0f G lines/s, %.2f TB/s of useful bytes\n", nv * 16, ms * 1e3, n,
                   n / (ms * 1e-3) / 1e9, double(n) * nv * 16 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
