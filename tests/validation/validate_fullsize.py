"""Full-size parity record (run on the GPU box): reddit-syn, 602 -> 128 -> 41, K epochs of
train + validation on the HIP path against the CPU checker with the reference's dropout decisions
replayed (HOST_MASKS), then the test split.  Uses the reference's own objects when
oracle/_ref/libref.so travelled with the repo, else the pinned C restatement.  Writes one JSON.

    python tests/validation/validate_fullsize.py [epochs=10] [out.json]
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from cuda_gcn_amd import datagen
from cuda_gcn_amd.model import HipGCNModel, HOST_MASKS
from oracle.pyoracle import Oracle, Ref


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    ds = datagen.make_dataset("reddit-syn")
    chk = Ref() if Ref.available() else Oracle()
    kind = "reference objects (oracle/_ref/libref.so)" if Ref.available() else "oracle/gcn_oracle.c"
    om = chk.model(ds, seed_time=3, hidden_dim=128, dropout=0.5)
    m = HipGCNModel(ds, seed=3, flags=HOST_MASKS, hidden_dim=128, dropout=0.5, epochs=K)
    rows = []
    t_cpu = t_gpu = 0.0
    for e in range(K):
        t0 = time.perf_counter(); got = m.train_epoch() + m.eval(2); t_gpu += time.perf_counter() - t0
        t0 = time.perf_counter(); want = om.train_epoch() + om.eval(2); t_cpu += time.perf_counter() - t0
        rows.append(dict(epoch=e + 1, hip=[float(x) for x in got], cpu=[float(x) for x in want]))
        print(f"epoch {e + 1}: hip {got}  cpu {want}", flush=True)
    gt, wt = m.eval(3), om.eval(3)
    g = np.array([r["hip"] for r in rows]); w = np.array([r["cpu"] for r in rows])
    res = dict(workload="reddit-syn 232965 nodes, 23446803 stored edges, 602->128->41, dropout 0.5 (reference RNG stream replayed), seed 3",
               checker=kind, epochs=K,
               max_abs_diff=dict(train_loss=float(np.abs(g[:, 0] - w[:, 0]).max()), train_acc=float(np.abs(g[:, 1] - w[:, 1]).max()),
                                 val_loss=float(np.abs(g[:, 2] - w[:, 2]).max()), val_acc=float(np.abs(g[:, 3] - w[:, 3]).max()),
                                 test_loss=abs(gt[0] - wt[0]), test_acc=abs(gt[1] - wt[1])),
               tolerance="|dloss| <= 2e-4 (epochs 1-10), |dacc| <= 2 / labelled rows",
               test=dict(hip=[float(x) for x in gt], cpu=[float(x) for x in wt]),
               seconds=dict(cpu_total=round(t_cpu, 1), hip_total_incl_host_mask_replay=round(t_gpu, 1)), trace=rows)
    print(json.dumps(res["max_abs_diff"]))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)
    m.close(); om.close()


if __name__ == "__main__":
    main()
