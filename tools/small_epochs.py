#!/usr/bin/env python3
"""N epochs (train + validation) of a small dataset at the reference's defaults (hidden 16, dropout 0.5), for profiling:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_epochs.py pubmed-syn 300
Prints the HIP-side wall time per epoch (run_epochs: no host synchronisation inside)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen  # noqa: E402
from cuda_gcn_amd.model import HipGCNModel  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pubmed-syn"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 16
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ds = datagen.make_dataset(name)
m = HipGCNModel(ds, seed=1, flags=flags, hidden_dim=hidden, dropout=0.5, epochs=n + 40)
m.run_epochs(20, want_trace=False)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    m.run_epochs(n, want_trace=False)
    best = min(best, time.perf_counter() - t0)
print(f"{name} hidden {hidden}: {1e6 * best / n:.1f} us per epoch (train + validation), best of 3 x {n} epochs")
m.close()
