"""2000 full-size epochs on one stream and with the validation lane: sustained rate and bit-identical traces.
Run on the GPU box:  python tests/validation/soak_2000_epochs.py   (measured: 283.5 and 297.7 epochs/s, identical)"""
import sys; sys.path.insert(0, '.')
import numpy as np, os, time
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, EVAL_LANE
ds = datagen.make_dataset("reddit-syn")
out = {}
# third leg: the opt-in in-launch segment sum (GCNHIP_GS_FOLD, graphsum.hip) — 2000 epochs x 5 aggregations x 10 K cross-XCD
# hand-offs each; one stale partial anywhere would change a bit of the trace
for name, fl in (("one", 0), ("lane", EVAL_LANE), ("lane+fold", EVAL_LANE)):
    os.environ.pop("GCNHIP_GS_FOLD", None)
    if name.endswith("fold"):
        os.environ["GCNHIP_GS_FOLD"] = "1"
    m = HipGCNModel(ds, seed=3, flags=fl, hidden_dim=128, dropout=0.5, epochs=2100)
    t0 = time.time(); tr = m.run_epochs(2000); dt = time.time() - t0
    print(name, "2000 epochs in %.2f s = %.1f epochs/s" % (dt, 2000 / dt), "finite", bool(np.isfinite(tr).all()), tr[-1], flush=True)
    out[name] = tr; m.close()
os.environ.pop("GCNHIP_GS_FOLD", None)
print("bit-identical:", np.array_equal(out["one"].view(np.uint32), out["lane"].view(np.uint32)),
      "with the in-launch segment sum:", np.array_equal(out["lane"].view(np.uint32), out["lane+fold"].view(np.uint32)))
