#!/bin/bash
# Counters of the kernels of a stand-alone HIP program (run on the GPU box):
#     PMC_MATCH=bf16x3,persist tools/pmc_exe.sh <out_dir> build/gemm_bf16x3 [args]
# One rocprofv3 run per counter group (--pmc with --kernel-trace only); prints a JSON summary, median per launch and kernel.
set -e
OUT=$(realpath -m "$1"); shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
EXE="$R/$1"; shift
export PMC_MATCH=${PMC_MATCH:-bf16x3}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- "$EXE" "$@" > "$OUT/p$i.log" 2>&1 || echo "FAILED pass $i ($C)"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys, collections
out = sys.argv[1]
match = os.environ["PMC_MATCH"].split(",")
want = lambda k: any(m in k for m in match)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
nm = lambda k: k.split("(")[0].replace("void ", "")
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if want(r["Kernel_Name"]): agg[nm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if want(r["Kernel_Name"]): dur[nm(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k, d in agg.items():
    e = {c: statistics.median(v) for c, v in d.items()}
    e["launches"] = max(len(v) for v in d.values())
    if dur.get(k):
        e["median_duration_us_under_pmc"] = statistics.median(dur[k]) / 1e3
        if e.get("GRBM_GUI_ACTIVE"): e["clock_GHz_from_GRBM_GUI_ACTIVE"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["median_duration_us_under_pmc"] * 1e3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_pipe_busy_share"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * e["SQ_BUSY_CU_CYCLES"])
    if e.get("SQ_WAVE_CYCLES") and e.get("SQ_WAIT_ANY") is not None:
        e["wave_cycles_waiting_share"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
    res[k] = e
print(json.dumps(res, indent=1, sort_keys=True))
PY
