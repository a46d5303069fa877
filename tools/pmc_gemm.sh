#!/bin/bash
# SQ counters of the dense first-layer GEMM kernels (run on the GPU box):  tools/pmc_gemm.sh <out_dir>
# One rocprofv3 run per counter group, --pmc with --kernel-trace only; prints a JSON summary (median per launch).
# Other kernels: PMC_PROG="tools/bench_ops.py reddit-syn 128 small" PMC_MATCH="rowstream,atb,xent" tools/pmc_gemm.sh <out_dir>
set -e
OUT=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
PROG="$R/${PMC_PROG:-tools/bench_gemm.py}"
export PMC_MATCH=${PMC_MATCH:-t128,persist}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 $PROG > "$OUT/p$i.log" 2>&1 || echo "FAILED pass $i"
done
GCN_REPO="$R" python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys, collections
out = sys.argv[1]
match = os.environ["PMC_MATCH"].split(",")
want = lambda k: any(m in k for m in match)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not want(k): continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want(k): dur[k.split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k, d in agg.items():
    e = {c: statistics.median(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if dur.get(k): e["median_duration_us_under_pmc"] = statistics.median(dur[k]) / 1e3
    # every CU has 4 SIMDs with one MFMA pipe each: busy share of the pipes while the kernel's CUs were busy
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_pipe_busy_share"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])
    res[k] = e
sys.path.insert(0, os.environ["GCN_REPO"])
from cuda_gcn_amd.provenance import source_sha
res["_meta"] = {"commit": os.environ.get("GCN_COMMIT"), "what": os.environ.get("PMC_PROG", "tools/bench_gemm.py") + " (reddit-syn shapes)", "counters": "median per launch; one rocprofv3 --pmc pass per counter group",
                "sources": source_sha(os.environ.get("PMC_SOURCES", "dense_bf16x3.h,bf16x3_split.h,dense_persist.h,class_bf16x3.h").split(","))}
print(json.dumps(res, indent=1, sort_keys=True))
PY
