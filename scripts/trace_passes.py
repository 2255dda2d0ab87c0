"""Per-kernel average durations of a rocprofv3 --kernel-trace of bench.py, split by the bench's passes: the timed pass
and the profiled pass (three frames in flight, kernels share the GPU) and the last pass (frames one after the other,
each kernel alone on the GPU - the duration bench.py's roofline uses).  usage: trace_passes.py <kernel_trace.csv> [steps]"""
import collections, csv, sys
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k in ("k_tile", "k_setup", "k_fill", "k_ascii_plane", "k_binscan"):
        if k in n:
            by[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-14s %8s %12s %14s %14s %14s" % ("kernel", "launches", "all (us)", "timed pass", "profiled pass", "serial pass"))
for k, v in sorted(by.items()):
    v.sort()
    d = [(e - s) / 1e3 for s, e in v]
    n = len(d)
    first = n - 2 * steps  # warm-up and timed frames
    print("%-14s %8d %12.2f %14.2f %14.2f %14.2f" % (k, n, sum(d) / n, sum(d[:first]) / first, sum(d[first:first + steps]) / steps, sum(d[-steps:]) / steps))
