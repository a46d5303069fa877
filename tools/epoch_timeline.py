"""One epoch of a kernel trace as a table (start, duration, stream/queue, kernel), cut between two Adam launches.

    rocprofv3 --kernel-trace --output-format csv -d out -o ks -- python3 bench.py --steps 10 --warmup 3 --bursts 0 --no-cpu-baseline --no-extras
    python tools/epoch_timeline.py out/**/ks_kernel_trace.csv [epoch index from the end, default 3]
"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k = sorted(({"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "q": r.get("Queue_Id", "?"), "st": r.get("Stream_Id", "?")}
            for r in rows), key=lambda x: x["s"])
adam = [i for i, x in enumerate(k) if x["name"].startswith("adam_kernel")]
# epochs of the timed region with the validation lane: between consecutive Adam launches
spans = [(adam[i], adam[i + 1]) for i in range(len(adam) - 1)]
# keep those whose launch count equals the most common one (drops the timers pass / train-only epochs)
from collections import Counter
common = Counter(b - a for a, b in spans).most_common(1)[0][0]
spans = [sp for sp in spans if sp[1] - sp[0] == common]
two = [sp for sp in spans if len({(x["q"], x["st"]) for x in k[sp[0] + 1:sp[1] + 1]}) > 1]     # epochs that use the second stream
if two:
    spans = two
queues = {}
def qid(x):
    key = (x["q"], x["st"])
    if key not in queues:
        queues[key] = len(queues) + 1
    return queues[key]
for a, b in spans[-back - 2:-back + 1]:
    seg = k[a + 1:b + 1]
    t0 = k[a]["e"]
    wall = (seg[-1]["e"] - t0) / 1e3
    ev = sorted([(x["s"], 1) for x in seg] + [(x["e"], -1) for x in seg])
    busy, depth, last = 0, 0, None
    for t, d in ev:
        if depth > 0:
            busy += t - last
        depth += d; last = t
    print(f"# epoch ending at launch {b}: wall {wall:.1f} us, some kernel running {busy / 1e3:.1f} us, sum of kernel durations {sum(x['e'] - x['s'] for x in seg) / 1e3:.1f} us, {len(seg)} launches")
a, b = spans[-back]
seg = k[a + 1:b + 1]
t0 = k[a]["e"]
print("   start      dur  stream  kernel")
for x in seg:
    print(f"{(x['s'] - t0) / 1e3:8.1f} {(x['e'] - x['s']) / 1e3:8.1f} {qid(x):7d}  {x['name'][:90]}")
