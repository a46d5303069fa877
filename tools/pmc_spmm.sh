#!/bin/bash
# PMC passes for the sparse SparseMatmul kernels (run on the GPU box through gpurun):
#   tools/pmc_spmm.sh <out_dir> [bench_spmm.py arguments, default: --only big --iters 5]
# One rocprofv3 run per counter group, --pmc combined with --kernel-trace only (MI355X_MICROARCH.md §rocprofv3 PMC slots).
set -e
OUT=$(realpath -m "$1"); shift
ARGS=${@:---only big --iters 5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/bench_spmm.py" $ARGS > "$OUT/p$i.log" 2>&1 || echo "FAILED pass $i ($C)"
done
python3 "$R/tools/pmc_summary.py" "$OUT" "${GCN_COMMIT:-unknown}" "tools/bench_spmm.py $ARGS" spmm > "$OUT/summary.json"
cat "$OUT/summary.json"
