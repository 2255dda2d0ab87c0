"""Per-kernel average durations of a rocprofv3 --kernel-trace of bench.py, split by the bench's passes: the timed rounds
(three frames in flight, kernels share the GPU), the pass bracketed by HIP events with frames overlapped, and the pass
with frames one after the other (each kernel alone on the GPU - the duration bench.py's roofline uses).  After those the
bench runs its latency and readback legs (blocking frames), which are left out.
usage: trace_passes.py <kernel_trace.csv> [frames per profiled pass = 300] [frames after the serial pass = 293]"""
import collections, csv, sys
n_prof = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 200 + 3 * 31
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("k_tile", "k_setup", "k_fill", "k_ascii", "k_binscan"):
        if k in n:
            by[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-14s %8s %12s %14s %14s %14s" % ("kernel", "launches", "all (us)", "timed rounds", "profiled pass", "serial pass"))
for k, v in sorted(by.items()):
    v.sort()
    d = [(e - s) / 1e3 for s, e in v]
    n = len(d)
    t = tail + (4 * 21 if k == "k_ascii" else 0)  # the ar_quantize legs at the end launch the smoothing kernel alone
    a, b, c = n - t - 2 * n_prof, n - t - n_prof, n - t
    print("%-14s %8d %12.2f %14.2f %14.2f %14.2f" % (k, n, sum(d) / n, sum(d[:a]) / a, sum(d[a:b]) / (b - a), sum(d[b:c]) / (c - b)))
