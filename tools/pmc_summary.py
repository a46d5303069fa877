"""Summarise the counter_collection CSVs of tools/pmc_graphsum.sh: per kernel, the median
counter value per launch, plus the HBM/fabric traffic per launch corrected as
MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE is in KiB and counts 128-byte requests at
64 bytes on gfx950 -> x2; WRITE_SIZE exact)."""
import collections, csv, glob, json, os, statistics, sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if "graphsum" not in k:
        continue
    d = {c: statistics.median(v) for c, v in cs.items()}
    d["launches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    out[k] = d
print(json.dumps(out, indent=1, sort_keys=True))
