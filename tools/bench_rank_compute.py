"""Per-rank COMPUTE time of the row-partitioned epoch, measured on one GPU (no 8-GPU box this round):
rank r of P is built exactly as in a real run (its row block, replicated first layer for P <= 4, the
rebuilt dH1, its own row schedule) but with no-op collectives (NULL_COMM), and K epochs are timed.
The slowest rank bounds the epoch from below; what a real run adds is the all-gathers' time that the
validation lane does not hide.  Losses are meaningless here (gather buffers are never filled).

    python tools/bench_rank_compute.py [dataset] [hidden] [P ...]        RANK_FLAGS=<int>: extra HipGCN flags, e.g. 1048576 =
    HIPGCN_OVERLAP_EXCHANGE (the aggregations then run as two launches over the cut operators: what the cut itself costs)
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime in the process)
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, NULL_COMM, NO_EVAL_LANE, TIMERS


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit-syn"
    hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    worlds = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
    ds = datagen.make_dataset(name)
    out = {}
    for P in worlds:
        per_rank = []
        for r in ([int(os.environ['RANK_ONLY'])] if os.environ.get('RANK_ONLY') else range(P)):    # RANK_ONLY: one rank under a profiler
            flags = NO_EVAL_LANE | (NULL_COMM if P > 1 else 0) | (TIMERS if os.environ.get("RANK_TIMERS") else 0) | int(os.environ.get("RANK_FLAGS", "0"))
            m = HipGCNModel(ds, seed=1, flags=flags, rank=r, world=P, hidden_dim=hidden, dropout=0.5, epochs=40)
            m.run_epochs(3, want_trace=False)
            m.sync()
            m.timers_reset()
            t0 = time.perf_counter()
            n_ep = int(os.environ.get('RANK_EPOCHS', '10'))
            m.run_epochs(n_ep, want_trace=False)
            m.sync()
            ms = 1e3 * (time.perf_counter() - t0) / n_ep
            info = m.info()
            if os.environ.get("RANK_TIMERS") and r == 0:
                bd = {}
                for nm in ("spmatmul_fw", "spmatmul_bw", "graphsum_fw", "graphsum_bw", "matmul_fw", "matmul_bw", "loss_fw", "adam", "comm"):
                    sec, n = m.timer(nm)
                    if n:
                        bd[nm] = round(1e3 * sec / n_ep, 4)
                print(f"  P={P} rank 0 timers (ms per epoch): {bd}  sum={sum(bd.values()):.3f}", flush=True)
            per_rank.append(dict(rank=r, ms=round(ms, 3), rows=info["local_rows"], edges=info["local_edges"], schedule=m.schedule()))
            m.close()                      # every rank is timed: blocks are balanced by EDGES, and the first-layer GEMM scales with ROWS
        out[P] = per_rank
        worst = max(x["ms"] for x in per_rank)
        print(f"P={P}: slowest rank {worst:.3f} ms per epoch (compute only) -> ceiling {1e3 / worst:.0f} epochs/s; {per_rank}", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
