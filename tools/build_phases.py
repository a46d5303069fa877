import sys, os
sys.path.insert(0, os.getcwd())
from cuda_gcn_amd import clirun, datagen
for name, hidden, ep in (("reddit-syn", 128, 10), ("rmat-22-256", 128, 5), ("pubmed-syn", 16, 10)):
    ds = datagen.make_dataset(name)
    r = clirun.run_on_dataset(ds, hidden=hidden, epochs=ep, env={"GCN_SEED": "1", "HIPGCN_VERBOSE": "1"})
    print("==", name, "load", r.get("load_s"), "build", r.get("model_build_s"), "process", round(r["process_wall_s"], 2))
    print("\n".join(l for l in r["stderr_tail"].splitlines() if "build" in l or "loaded" in l))
