"""Time model of the row-partitioned epoch on 1/2/4/8 MI355X of one node: what the first SCALE record should
be read against.  Host only.  Nothing here has been measured on more than one GPU — the inputs are:

  * per-rank COMPUTE time (ms per epoch, slowest rank) measured on one GPU with no-op collectives
    (tools/bench_rank_compute.py -> profiles/*rank_compute*): --compute "1:3.53,2:2.21,4:1.48,8:0.78";
  * the exchanges of one epoch (train + validation) and their row widths, from how HipGCN wires the model:
        P <= 4 with an all-gather plan (first layer replicated): Z0, H1>0 bits, dZ, dZ0, validation Z0
        otherwise: + H0 at the hidden width;
    rows per exchange from the exchange plan of the neediest rank (host/partition.h);
  * xGMI: 153 GB/s per link and direction, one link per peer (MI355X_MICROARCH / SURVEY §5).  `direct`: every peer's block
    arrives over its own link at the same time, time = bytes_from_one_peer / 153 GB/s.  `ring`: bytes_total / 153 GB/s;
  * a fixed cost per collective (launch + rendezvous), --latency-us (default 20, an assumption);
  * overlap (HIPGCN_OVERLAP_EXCHANGE): an aggregation's exchange runs beside its own-column edges, whose share of the
    rank's edges is counted from the partition; the bits exchange is consumed an aggregation later (fully hidden); the dZ0
    exchange runs beside dW2 and the own rows' dH1 (taken as 0.1 ms / P).  The cut itself costs compute (two launches per
    aggregation, the output rows written and re-read): --compute-overlap takes the per-rank compute measured with the flag.

    python tools/comm_model.py reddit-syn --compute 1:3.37,2:2.21,4:1.48,8:0.78 --agg-ms 0.75,0.28,0.27,0.10
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen, model  # noqa: E402

LINK_GBPS = 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dataset")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--compute", required=True, help="P:ms,... per-rank compute per epoch (slowest rank, no-op collectives)")
    ap.add_argument("--agg-ms", default="0.75,0.28,0.27,0.10",
                    help="one-GPU times of the aggregations that follow an exchange: hidden fwd, class fwd (train), class bwd, class fwd (validation)")
    ap.add_argument("--latency-us", type=float, default=20.0,
                    help="device time per collective; pass the one-rank figure tools/rccl_latency.py measures (a lower bound on what a peer adds)")
    ap.add_argument("--latency-source", default="assumed", help="text for the report line: where --latency-us came from")
    ap.add_argument("--compute-overlap", default="", help="P:ms,... the same measurement with HIPGCN_OVERLAP_EXCHANGE (RANK_FLAGS=1048576): "
                                                          "the aggregations run as two launches over the cut operators, which has a cost of its own")
    a = ap.parse_args()
    comp = {int(k): float(v) for k, v in (kv.split(":") for kv in a.compute.split(","))}
    comp_ovl = {int(k): float(v) for k, v in (kv.split(":") for kv in a.compute_overlap.split(","))} if a.compute_overlap else {}
    agg_h, agg_c, agg_cb, agg_cv = (float(x) for x in a.agg_ms.split(","))
    if a.dataset.startswith("rmat"):
        gp, gi = datagen.rmat_graph(int(a.dataset.split("-")[1]))
    else:
        ds = datagen.make_dataset(a.dataset)
        gp, gi = ds["g_indptr"], ds["g_indices"]
    N = gp.size - 1
    ldH, ldC, wpr = (a.hidden + 15) // 16 * 16, 48, (a.hidden + 31) // 32
    lat = a.latency_us * 1e-3
    print(f"{a.dataset}: N={N}, stored edges={gi.size}; link {LINK_GBPS:.0f} GB/s per peer, {a.latency_us:.1f} us per collective ({a.latency_source})")
    print(f"{'P':>2} {'plan':>9} {'rows/exch':>10} {'MB/epoch':>9} {'own cols':>8} | {'compute':>7} | {'comm direct':>11} {'ring':>6} | "
          f"{'no overlap':>22} | {'overlap':>22}")
    base = comp[1]
    print(f"{1:>2} {'-':>9} {0:>10} {0:>9} {'100 %':>8} | {base:7.2f} | {0:11.2f} {0:6.2f} | {1e3 / base:8.0f} epochs/s  1.00x | {1e3 / base:8.0f} epochs/s  1.00x")
    for P in sorted(k for k in comp if k > 1):
        start, _ = model.partition(gp, P)
        worst = None
        for r in range(P):
            p = model.exchange_plan(gp, gi, P, r, 0)
            recv = p["recv_rows"].size if p["halo"] else (P - 1) * p["rows_max"]
            if worst is None or recv > worst[0]:
                r0, r1 = int(start[r]), int(start[r + 1])
                cols = gi[gp[r0]:gp[r1]]
                own = float(((cols >= r0) & (cols < r1)).mean())
                worst = (recv, p["halo"], own, p["rows_max"])
        rows, halo, own, rows_max = worst
        replicated = (not halo) and P <= 4
        # (name, words per row, one-GPU time of what can run beside it with the overlap flag)
        exch = [("Z0", ldC, agg_c / P * own), ("bits", wpr, 1e9), ("dZ", ldC, agg_cb / P * own), ("dZ0", ldC, 0.1 / P), ("Z0 val", ldC, agg_cv / P * own)]
        if not replicated:
            exch.insert(0, ("H0", ldH, agg_h / P * own))
        n_coll = len(exch) + 2                               # + gradient all-reduce + the validation scalars
        mb = sum(w for _, w, _ in exch) * 4 * rows / 1e6
        peers = P - 1
        t_direct = [w * 4 * rows / peers / (LINK_GBPS * 1e6) for _, w, _ in exch]     # ms: one peer's share over its own link
        t_ring = [w * 4 * rows / (LINK_GBPS * 1e6) for _, w, _ in exch]
        comm_d, comm_r = sum(t_direct) + n_coll * lat, sum(t_ring) + n_coll * lat
        hidden = sum(min(t, h) for t, (_, _, h) in zip(t_direct, exch))
        t_plain, t_ovl = comp[P] + comm_d, comp_ovl.get(P, comp[P]) + comm_d - hidden
        print(f"{P:>2} {'halo' if halo else 'allgather':>9} {rows:>10} {mb:9.1f} {own:8.1%} | {comp[P]:7.2f} | {comm_d:11.2f} {comm_r:6.2f} | "
              f"{1e3 / t_plain:8.0f} epochs/s {base / t_plain:5.2f}x | {1e3 / t_ovl:8.0f} epochs/s {base / t_ovl:5.2f}x")


if __name__ == "__main__":
    main()
