import sys; sys.path.insert(0, '.')
import torch, numpy as np
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel
ds = datagen.make_dataset("reddit-syn")
m = HipGCNModel(ds, seed=11, hidden_dim=128, dropout=0.5, epochs=400)
tr = m.run_epochs(400)
print("finite:", np.isfinite(tr).all(), "epoch 1", tr[0], "epoch 100", tr[99], "epoch 400", tr[-1], "test", m.eval(3))
m.close()
