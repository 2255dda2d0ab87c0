"""Writes build_variants/diverge.hip: the library source with counters of how evenly the lanes of a wavefront are loaded
in the per-lane raster of k_tile (diagnostic build only; never shipped).  Per call of raster_small: rows walked and
candidate cells tested by the lane, summed over the lanes, and the wavefront's maximum of each (what the wavefront
pays for, times 64).  ar_debug_divergence(out[6]) returns {calls, lanes, sum rows, sum of max rows, sum candidates,
sum of max candidates} since the library was loaded."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "scripts"))
from flatten_source import flatten
s = flatten()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
rep("namespace {\n\nconstexpr uint64_t kBackgroundKey", """__device__ unsigned long long g_diverge[6];
__device__ __forceinline__ void diverge_note(int rows, int cands) {
    // called by the lanes that rasterise; the others of the wavefront are idle for the duration
    const unsigned long long active = __ballot(1);
    int mr = rows, mc = cands, sr = rows, sc = cands;
    for (int off = 32; off >= 1; off >>= 1) {
        const int orow = __shfl_xor(mr, off), ocand = __shfl_xor(mc, off), osr = __shfl_xor(sr, off), osc = __shfl_xor(sc, off);
        const bool other = (active >> ((threadIdx.x & 63u) ^ off)) & 1ull;
        if (other) { mr = max(mr, orow); mc = max(mc, ocand); sr += osr; sc += osc; }
    }
    if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)active) - 1)) {
        atomicAdd(&g_diverge[0], 1ull); atomicAdd(&g_diverge[1], (unsigned long long)__popcll(active));
        atomicAdd(&g_diverge[2], (unsigned long long)sr); atomicAdd(&g_diverge[3], (unsigned long long)mr);
        atomicAdd(&g_diverge[4], (unsigned long long)sc); atomicAdd(&g_diverge[5], (unsigned long long)mc);
    }
}
namespace {

constexpr uint64_t kBackgroundKey""")
rep("    uint32_t clo = uint32_t(cand), chi = uint32_t(cand >> 32);\n    while (clo | chi) {\n        int k;\n        if (clo) { k = __ffs(int(clo)) - 1; clo &= clo - 1u; }\n        else { k = 31 + __ffs(int(chi)); chi &= chi - 1u; }\n        const int rr = r0 + (k >> ls)",
    "    diverge_note(r1 - r0 + 1, __popcll(cand));\n    uint32_t clo = uint32_t(cand), chi = uint32_t(cand >> 32);\n    while (clo | chi) {\n        int k;\n        if (clo) { k = __ffs(int(clo)) - 1; clo &= clo - 1u; }\n        else { k = 31 + __ffs(int(chi)); chi &= chi - 1u; }\n        const int rr = r0 + (k >> ls)")
s += '''
extern "C" int ar_debug_divergence(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diverge), 6 * 8) == hipSuccess ? 0 : -1;
}
'''
os.makedirs(os.path.join(root, "build_variants"), exist_ok=True)
open(os.path.join(root, "build_variants/diverge.hip"), "w").write(s)
