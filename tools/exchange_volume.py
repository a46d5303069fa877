"""Rows and bytes each rank receives per exchange and per epoch, ALLGATHER (round 1) vs the plan picked now
(host/partition.h), for the BASELINE multi-GPU configurations.  Host only.
    python tools/exchange_volume.py reddit-syn 128      python tools/exchange_volume.py rmat-22 128
Tables a rank completes per epoch (train + validation): H0 (hidden width; not exchanged when the first layer is
replicated, 2-4 GPUs with an ALLGATHER plan), Z0, dZ and dZ0 (class width, ld 48), the H1 > 0 bits (hidden/32 words),
and Z0 of the validation forward.  Round 1 also gathered H0 for the validation forward; aggregate-first evaluation
(A^.X built once) removed that one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen, model

name = sys.argv[1]
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if name.startswith("rmat"):
    gp, gi = datagen.rmat_graph(int(name.split("-")[1]))
else:
    ds = datagen.make_dataset(name)
    gp, gi = ds["g_indptr"], ds["g_indices"]
N = gp.size - 1
ldH, ldC, wpr = (H + 15) // 16 * 16, 48, (H + 31) // 32
print(f"{name}: N={N} stored edges={gi.size}")
for P in (2, 4, 8):
    worst = None
    for r in range(P):
        p = model.exchange_plan(gp, gi, P, r, 0)
        recv_halo = p["recv_rows"].size if p["halo"] else (P - 1) * p["rows_max"]
        recv_ag = (P - 1) * p["rows_max"]
        if worst is None or recv_halo > worst[0]:
            worst = (recv_halo, recv_ag, p["halo"], p["halo_share"])
    rh, ra, halo, share = worst
    replicated = (not halo) and P <= 4
    words = (0 if replicated else ldH) + 4 * ldC + wpr
    words_r1 = (0 if P <= 4 else 2 * ldH) + 4 * ldC + wpr
    print(f"  P={P}: plan={'halo' if halo else 'allgather'} (neediest rank reads {share:.1%} of the remote rows); "
          f"rows received per exchange: {rh} (all-gather of padded blocks: {ra}); per epoch {rh * words * 4 / 1e6:.1f} MB "
          f"(round 1: {ra * words_r1 * 4 / 1e6:.1f} MB)")
