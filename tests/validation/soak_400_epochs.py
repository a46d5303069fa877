"""400 epochs of the full-size workload on one GPU, three ways: one stream replaying the captured hipGraph epoch, the
validation lane on a second stream (bench.py's default with one GPU), and the opt-in two-stream backward pipeline.
Stream ordering is what this checks at full size: the three traces must agree to the bit, epoch by epoch.
Run on the GPU box:  python tests/validation/soak_400_epochs.py"""
import sys; sys.path.insert(0, '.')
import numpy as np
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, EVAL_LANE, BWD_PIPELINE

ds = datagen.make_dataset("reddit-syn")
traces = {}
for name, flags in (("one stream", 0), ("validation lane", EVAL_LANE), ("backward pipeline + lane", EVAL_LANE | BWD_PIPELINE)):
    m = HipGCNModel(ds, seed=11, flags=flags, hidden_dim=128, dropout=0.5, epochs=400)
    tr = m.run_epochs(400)
    test = m.eval(3)
    print(f"{name}: finite {bool(np.isfinite(tr).all())}; epoch 1 {tr[0]}; epoch 100 {tr[99]}; epoch 400 {tr[-1]}; test {test}", flush=True)
    traces[name] = (tr, test)
    m.close()
ref = traces["one stream"]
for name, (tr, test) in traces.items():
    same = np.array_equal(tr.view(np.uint32), ref[0].view(np.uint32)) and test == ref[1]
    print(f"{name}: bit-identical to one stream: {same}")
    assert same, name
print("ok")
