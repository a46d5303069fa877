#!/bin/bash
# PMC passes for the GraphSum kernels (run on the GPU box through gpurun):
#   tools/pmc_graphsum.sh <out_dir> [dataset] [hidden]
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md §rocprofv3 PMC slots); --pmc is combined with --kernel-trace only.
set -e
OUT=$1; DS=${2:-reddit-syn}; H=${3:-128}; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr " " "_")
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$N" -- python3 "$R/tools/bench_ops.py" $DS $H graphsum $EXTRA > "$OUT/$N.log" 2>&1 || echo "FAILED $N"
done
python3 "$R/tools/pmc_summary.py" "$OUT" "${GCN_COMMIT:-unknown}" "tools/bench_ops.py $DS $H graphsum $EXTRA" > "$OUT/summary.json"
cat "$OUT/summary.json"
