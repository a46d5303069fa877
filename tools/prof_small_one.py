import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, NO_GRAPH
ds = datagen.make_dataset("cora-syn")
m = HipGCNModel(ds, seed=1, flags=NO_GRAPH, hidden_dim=16, dropout=0.5, epochs=700)
m.run_epochs(50, want_trace=False)
t0 = time.perf_counter(); m.run_epochs(500, want_trace=False); dt = time.perf_counter() - t0
print(f"cora-syn: {1e6 * dt / 500:.1f} us/epoch")
m.close()
