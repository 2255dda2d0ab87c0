"""Writes build_variants/stamps.hip: the kernel source with wall-clock stamps at the phase boundaries of
k_tile (diagnostic build only; never shipped).  Thread 0 of each tile records the 100 MHz clock after
the barrier that ends each phase."""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "scripts"))
from flatten_source import flatten
s = flatten()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:60])
    s = s.replace(a, b)
rep("namespace {\n\nconstexpr uint64_t kBackgroundKey", "__device__ unsigned long long g_stamps[8192 * 8];\n#define STAMP(k) do { if (tid == 0) g_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)\nnamespace {\n\nconstexpr uint64_t kBackgroundKey")

rep("    for (int c = int(tid); c < kTileCells; c += kTileThreads) keys[c] = kBackgroundKey;\n    if (tid == 0) { n_big = 0; n_cov = 0; n_marked = 0; }", "    STAMP(0);\n    for (int c = int(tid); c < kTileCells; c += kTileThreads) keys[c] = kBackgroundKey;\n    if (tid == 0) { n_big = 0; n_cov = 0; n_marked = 0; }")
rep("        if (nl + kTileThreads * kFilterBatch > uint32_t(kTileListMax) || last) {\n", "        if (nl + kTileThreads * kFilterBatch > uint32_t(kTileListMax) || last) {\n            STAMP(1); if (tid == 0) g_stamps[blockIdx.x * 8 + 7] = (unsigned long long)nl << 40 | (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32 | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);\n")
rep("            __syncthreads();\n            // sweep: the marked entries are collected", "            __syncthreads();\n            STAMP(2);\n            // sweep: the marked entries are collected")
rep("            nl = 0;\n        }\n        if (last) break;", "            STAMP(3);\n            nl = 0;\n        }\n        if (last) break;")
rep("    __syncthreads();\n    const uint32_t nc = n_cov;", "    __syncthreads();\n    STAMP(4);\n    const uint32_t nc = n_cov;")
rep("    __syncthreads();\n\n    // colour and ramp index of a resolved key", "    __syncthreads();\n    STAMP(5);\n\n    // colour and ramp index of a resolved key")
rep("            reinterpret_cast<uint32_t *>(ridx)[cell >> 2] = ri4;\n        }\n        return;", "            reinterpret_cast<uint32_t *>(ridx)[cell >> 2] = ri4;\n        }\n        STAMP(6);\n        return;")
# end of k_tile: find the closing of the output loop
i = s.index("// ascii_stage.h")
j = s.rindex("}\n}\n", 0, i)
s = s[:j] + "}\n    STAMP(6);\n}\n" + s[j + 4:]
s += '\nextern "C" int ar_debug_stamps(unsigned long long *out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), size_t(n) * 8) == hipSuccess ? 0 : -1; }\n'
os.makedirs(os.path.join(root, "build_variants"), exist_ok=True)
open(os.path.join(root, "build_variants/stamps.hip"), "w").write(s)
