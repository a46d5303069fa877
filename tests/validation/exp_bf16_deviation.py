"""How far does the opt-in bf16 table storage move the training trace from the f32 reference path?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, HOST_MASKS, BF16_TABLES
from oracle.pyoracle import Oracle

o = Oracle()
for name, hidden, epochs in (("cora-syn", 16, 100), ("pubmed-syn", 16, 100), ("reddit-mini", 128, 30)):
    ds = datagen.make_dataset(name)
    om = o.model(ds, seed_time=5, hidden_dim=hidden, dropout=0.5)
    a = HipGCNModel(ds, seed=5, flags=HOST_MASKS, hidden_dim=hidden, dropout=0.5, epochs=epochs)
    b = HipGCNModel(ds, seed=5, flags=HOST_MASKS | BF16_TABLES, hidden_dim=hidden, dropout=0.5, epochs=epochs)
    W = []; A = []; B = []
    for e in range(epochs):
        W.append(om.train_epoch() + om.eval(2)); A.append(a.train_epoch() + a.eval(2)); B.append(b.train_epoch() + b.eval(2))
    W, A, B = (np.array(t, np.float64) for t in (W, A, B))
    print(f"{name} h={hidden} {epochs} epochs: f32 path vs oracle: max |dloss| {np.abs(A[:, [0, 2]] - W[:, [0, 2]]).max():.2e}, max |dacc| {np.abs(A[:, [1, 3]] - W[:, [1, 3]]).max():.2e}")
    print(f"    bf16 tables vs oracle: max |dloss| {np.abs(B[:, [0, 2]] - W[:, [0, 2]]).max():.2e} (rel {np.abs(B[:, [0, 2]] / W[:, [0, 2]] - 1).max():.2e}), max |dacc| {np.abs(B[:, [1, 3]] - W[:, [1, 3]]).max():.2e}; "
          f"final val acc {B[-1, 3]:.4f} vs {W[-1, 3]:.4f}; test {b.eval(3)} vs {om.eval(3)}", flush=True)
    a.close(); b.close(); om.close()
