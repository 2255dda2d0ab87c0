"""Writes build_variants/timeline.hip: the library source with a device-side log of when each of a frame's five
kernels starts and ends (diagnostic build only; never shipped).  The first workgroup writes the 100 MHz wall clock at
entry, every eighth workgroup folds it into a maximum at exit, so the log shows how the kernels of
overlapped frames really interleave, without a profiler serialising or slowing the launches.
ar_debug_timeline_reset() / ar_debug_timeline_read(out, n_frames) are exported by the variant."""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "scripts"))
from flatten_source import flatten
s = flatten()
def rep(a, b, count=1):
    global s
    assert s.count(a) == count, (s.count(a), a[:60])
    s = s.replace(a, b)
rep("namespace {\n\nconstexpr uint64_t kBackgroundKey", """__device__ unsigned long long g_timeline[4096 * 5 * 2];
struct TimelineMark {
    unsigned long long *e;
    __device__ TimelineMark(int k, unsigned frame) : e(g_timeline + ((frame & 4095u) * 5 + k) * 2) {
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) *e = (unsigned long long)wall_clock64();
    }
    __device__ ~TimelineMark() {  // (a sample of the workgroups: every one of them would queue up on the one address)
        if (threadIdx.x == 0 && ((blockIdx.x + blockIdx.y * gridDim.x) & 7u) == 0u) atomicMax(e + 1, (unsigned long long)wall_clock64());
    }
};
namespace {

constexpr uint64_t kBackgroundKey""")
# (every kernel of the frame binds `fp` - its own parameters, or the frame's of a launch over several - in its first line)
bind = "const FrameParams &fp = BATCH ? fps[blockIdx.z] : fp_one;\n"
for k, name in enumerate(["k_setup", "k_binscan", "k_fill", "k_tile", "k_ascii_counts"]):
    i = s.index("void %s(" % name)
    j = s.index(bind, i) + len(bind)
    s = s[:j] + "    TimelineMark timeline_mark(%d, fp.frame);\n" % k + s[j:]
s += '''
extern "C" int ar_debug_timeline_reset() {
    static unsigned long long init[4096 * 5 * 2];
    for (int i = 0; i < 4096 * 5; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), init, sizeof init) == hipSuccess ? 0 : -1;
}
extern "C" int ar_debug_timeline_read(unsigned long long *out, int n_frames) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), size_t(n_frames) * 5 * 2 * 8) == hipSuccess ? 0 : -1;
}
'''
os.makedirs(os.path.join(root, "build_variants"), exist_ok=True)
open(os.path.join(root, "build_variants/timeline.hip"), "w").write(s)
