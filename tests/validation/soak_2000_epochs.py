"""2000 full-size epochs on one stream and with the validation lane: sustained rate and bit-identical traces; a third leg
with the reference's per-edge coefficients (EDGE_COEF) for the rate of the unfactored operator on the same box.
Run on the GPU box:  python tests/validation/soak_2000_epochs.py"""
import sys; sys.path.insert(0, '.')
import numpy as np, time
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, EVAL_LANE, EDGE_COEF
ds = datagen.make_dataset("reddit-syn")
out = {}
for name, fl in (("one stream", 0), ("lane", EVAL_LANE), ("lane, per-edge coefficients", EVAL_LANE | EDGE_COEF)):
    m = HipGCNModel(ds, seed=3, flags=fl, hidden_dim=128, dropout=0.5, epochs=2100)
    t0 = time.time(); tr = m.run_epochs(2000); dt = time.time() - t0
    print(name, "2000 epochs in %.2f s = %.1f epochs/s" % (dt, 2000 / dt), "finite", bool(np.isfinite(tr).all()), tr[-1], flush=True)
    out[name] = tr; m.close()
print("one stream vs lane bit-identical:", np.array_equal(out["one stream"].view(np.uint32), out["lane"].view(np.uint32)))
d = np.abs(out["lane"] - out["lane, per-edge coefficients"])
print("factored vs per-edge coefficients over 2000 epochs: max |d loss| train %.2e val %.2e, max |d acc| %.2e" % (d[:, 0].max(), d[:, 2].max(), max(d[:, 1].max(), d[:, 3].max())))
