"""Builds profiles/traffic.json (HBM bytes per launch and kernel) from two rocprofv3 --pmc passes.

usage: traffic_json.py <dir with pmc_fetch/ and pmc_write/> <out.json> [<dir with pmc_ea/ from scripts/pmc_l2.sh> [<dir with pmc_busy/ from scripts/pmc_busy.sh>]]
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes,
so the read side is doubled (verified here on k_setup, whose reads are a known 37 bytes per triangle);
for kernels whose reads are scattered 64-byte records the doubled figure is an upper bound.
"""
import collections, csv, glob, importlib, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
res = {"_frames_per_launch": int(os.environ.get("AR_CHAIN", "1") or 1),  # frames a launch of the counter runs held (scripts/pmc_run.py)
       "_note": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; _raw holds the counters in KiB",
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


def busy_pass(path):
    """scripts/pmc_busy.sh: SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE per dispatch, joined with the kernel
    trace of the same run.  GRBM_GUI_ACTIVE is summed over the eight XCDs and spans a fixed stretch around each kernel,
    so the engine clock under load is the slope of GUI_ACTIVE / 8 against the kernel's duration over the dispatches."""
    traces = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
    counters = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
    if not traces or not counters:
        return None
    dur = {}
    for row in csv.DictReader(open(traces[0])):
        dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    per = collections.defaultdict(dict)
    names = {}
    for row in csv.DictReader(open(counters[0])):
        per[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    pts = [(dur[d], c["GRBM_GUI_ACTIVE"] / 8.0) for d, c in per.items() if names[d].startswith("k_") and d in dur and "GRBM_GUI_ACTIVE" in c]
    n = len(pts)
    mx, my = sum(p[0] for p in pts) / n, sum(p[1] for p in pts) / n
    slope = sum((p[0] - mx) * (p[1] - my) for p in pts) / sum((p[0] - mx) ** 2 for p in pts)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, c in per.items():
        k = names[d]
        if not k.startswith("k_") or d not in dur:
            continue
        k = "k_ascii" if k.startswith("k_ascii") else k
        acc[k]["duration_ns"].append(dur[d])
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"):
            if name in c:
                acc[k][name].append(c[name])
    kernels = {k: {name: sum(v) / len(v) for name, v in d.items()} for k, d in sorted(acc.items())}
    return {"clock_ghz_under_load": slope, "fit_points": n, "kernels": kernels,
            "note": "per launch under counters: SQ_ACTIVE_INST_VALU in units of four cycles of one SIMD, SQ_BUSY_CU_CYCLES summed over "
                    "the 256 compute units, GRBM_GUI_ACTIVE summed over the 8 XCDs; clock = slope of GUI_ACTIVE / 8 over duration"}


if len(sys.argv) > 4:
    b = busy_pass(sys.argv[4])
    if b:
        res["_busy"] = b
res["_csrc_hash"] = importlib.import_module("ascii-renderer_amd.provenance").csrc_hash()
res["_csrc_hash_note"] = "fingerprint of ascii-renderer_amd/csrc when the counters were taken; bench.py drops these figures when the sources have changed since"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.startswith("k_")}))
