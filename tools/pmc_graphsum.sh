#!/bin/bash
# PMC passes for the GraphSum kernels (run on the GPU box through gpurun):
#   tools/pmc_graphsum.sh <out_dir> [dataset] [hidden] [mode of tools/bench_ops.py: only_h | only_c | only_c64 | only_split ...]
# The same three memory-side passes for any other program / kernels:
#   PMC_PROG="tools/bench_class.py 232965 10 noabl" PMC_MATCH="class_,gemm_,slab_reduce" tools/pmc_graphsum.sh <out_dir>
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md §rocprofv3 PMC slots); --pmc is combined with --kernel-trace only.
set -e
OUT=$1; DS=${2:-reddit-syn}; H=${3:-128}; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr " " "_")
  if [ -n "$PMC_PROG" ]; then
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$N" -- python3 $R/$PMC_PROG > "$OUT/$N.log" 2>&1 || echo "FAILED $N"
  else
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$N" -- python3 "$R/tools/bench_ops.py" $DS $H graphsum $EXTRA > "$OUT/$N.log" 2>&1 || echo "FAILED $N"
  fi
done
python3 "$R/tools/pmc_summary.py" "$OUT" "${GCN_COMMIT:-unknown}" "${PMC_PROG:-tools/bench_ops.py $DS $H graphsum $EXTRA}" "${PMC_MATCH:-graphsum}" > "$OUT/summary.json"
cat "$OUT/summary.json"
