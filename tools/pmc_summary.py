"""Summarise the counter_collection CSVs of tools/pmc_graphsum.sh: per kernel (full name, template arguments included),
the median counter value per launch, the kernel's median duration under the profiler (kernel_trace CSVs of the same
runs) and the HBM/fabric traffic per launch corrected as MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE is in KiB and
counts 128-byte requests at 64 bytes on gfx950 -> x2; WRITE_SIZE exact), with the cross-check of that correction the
guide asks for on one's own access pattern: TCC_MISS_sum x 128 B (every L2 miss fetches one 128-byte line) against
2 x FETCH_SIZE.  `_meta` ties the file to the tree it was taken on (bench.py refuses a profile whose kernel name or
duration does not match the launch it has just timed).

    python3 tools/pmc_summary.py <dir> [commit] [what was run] [kernel-name substrings, comma separated; default graphsum]
"""
import collections, csv, glob, json, os, statistics, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd.provenance import source_sha  # noqa: E402

root = sys.argv[1]
match = (sys.argv[4] if len(sys.argv) > 4 else "graphsum").split(",")      # kernel-name substrings, comma separated
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)


def name_of(r):
    return r["Kernel_Name"].split("(")[0].replace("void ", "")


for f in glob.glob(os.path.join(root, "*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[name_of(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(root, "*", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[name_of(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"_meta": {"commit": sys.argv[2] if len(sys.argv) > 2 else None, "what": sys.argv[3] if len(sys.argv) > 3 else None,
                 "counters": "median per launch; one rocprofv3 --pmc pass per counter group, --kernel-trace only",
                 # the kernel sources this file describes (PMC_SOURCES, comma separated; default: the aggregation): bench.py marks the
                 # file stale when one of them has changed (cuda_gcn_amd/provenance.py)
                 "sources": source_sha(os.environ.get("PMC_SOURCES", "graphsum.hip").split(","))}}
for k, cs in agg.items():
    if not any(m in k for m in match):
        continue
    d = {c: statistics.median(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if dur.get(k):
        d["median_duration_us_under_pmc"] = statistics.median(dur[k]) / 1e3
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if "FETCH_SIZE" in d:
            d["fetch_crosscheck_TCC_MISS_x128_over_2xFETCH_SIZE"] = d["TCC_MISS_sum"] * 128 / (2 * d["FETCH_SIZE"] * 1024)
    if d.get("SQ_BUSY_CU_CYCLES") and d.get("median_duration_us_under_pmc"):
        d["clock_GHz_from_busy_cu_cycles"] = d["SQ_BUSY_CU_CYCLES"] / 256.0 / (d["median_duration_us_under_pmc"] * 1e3)
    if d.get("SQ_WAVE_CYCLES") and d.get("SQ_WAIT_ANY") is not None:
        d["wave_cycles_waiting_share"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    out[k] = d
print(json.dumps(out, indent=1, sort_keys=True))
