"""µs/epoch of the launch-bound small graphs (Cora/Citeseer/Pubmed shapes, hidden 16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, NO_GRAPH
for flags in (NO_GRAPH, 0):
  print('hipGraph replay' if flags == 0 else 'eager launches')
  for name in ("cora-syn", "citeseer-syn", "pubmed-syn"):
      ds = datagen.make_dataset(name)
      m = HipGCNModel(ds, seed=1, flags=flags, hidden_dim=16, dropout=0.5, epochs=700)
      m.run_epochs(50, want_trace=False)
      t0 = time.perf_counter(); m.run_epochs(500, want_trace=False); dt = time.perf_counter() - t0
      print(f"{name}: {1e6 * dt / 500:.1f} us/epoch (train+val), {500 / dt:.0f} epochs/s", flush=True)
      m.close()
