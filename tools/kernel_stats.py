#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (CSV output): calls, total and average duration, share.
    python tools/kernel_stats.py <dir with *_kernel_trace.csv> [launches per epoch divisor]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
per = float(sys.argv[2]) if len(sys.argv) > 2 else None
d = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in d.values())
print(f"{'kernel':70s} {'calls':>7s} {'avg us':>9s} {'total ms':>9s} {'share':>6s}" + ("  us/epoch" if per else ""))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    line = f"{k[:70]:70s} {len(v):7d} {sum(v) / len(v) / 1e3:9.2f} {sum(v) / 1e6:9.2f} {100 * sum(v) / tot:5.1f}%"
    if per:
        line += f" {sum(v) / 1e3 / per:9.2f}"
    print(line)
print(f"{'all kernels':70s} {sum(len(v) for v in d.values()):7d} {'':9s} {tot / 1e6:9.2f}" + (f" {'':6s} {tot / 1e3 / per:9.2f}" if per else ""))
