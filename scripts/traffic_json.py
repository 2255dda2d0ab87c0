"""Builds profiles/traffic.json (HBM bytes per launch and kernel) from two rocprofv3 --pmc passes.

usage: traffic_json.py <dir with pmc_fetch/ and pmc_write/> <out.json> [<dir with pmc_ea/ from scripts/pmc_l2.sh>]
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes,
so the read side is doubled (verified here on k_setup, whose reads are a known 37 bytes per triangle);
for kernels whose reads are scattered 64-byte records the doubled figure is an upper bound.
"""
import collections, csv, glob, json, sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            acc[k].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


src, out = sys.argv[1], sys.argv[2]
fetch, write = per_kernel(src + "/pmc_fetch", "FETCH_SIZE"), per_kernel(src + "/pmc_write", "WRITE_SIZE")
res = {"_note": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; _raw holds the counters in KiB",
       "_raw": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("k_"):
        res["k_ascii" if k.startswith("k_ascii") else k] = int((2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024)
if len(sys.argv) > 3:
    rd = per_kernel(sys.argv[3] + "/pmc_ea", "TCC_EA0_RDREQ_sum")
    req = {}
    for k, v in sorted(rd.items()):
        if k.startswith("k_"):
            req["k_ascii" if k.startswith("k_ascii") else k] = int(v)
    res["_read_requests"] = req
    res["_read_requests_note"] = ("memory-side read requests per launch (TCC_EA0_RDREQ_sum); the scattered-read ceiling "
                                  "measured by scripts/micro/fetch_granularity.hip is 36e9 lines/s")
valu = per_kernel(src + "/pmc_inst", "SQ_INSTS_VALU")  # (present when the directory comes from scripts/profile_round.sh)
if valu:
    res["_valu_instructions"] = {("k_ascii" if k.startswith("k_ascii") else k): int(v) for k, v in sorted(valu.items()) if k.startswith("k_")}
    res["_valu_instructions_note"] = "wave-level vector instructions per launch (SQ_INSTS_VALU); one occupies a SIMD for four cycles"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.startswith("k_")}))
