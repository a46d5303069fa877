#!/bin/bash
# Round-5 measurement batch, part 2: memory-side counters of the class-width table layouts and of the class-layer kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5i
mkdir -p $O
export GCN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
cd $R
for mode in only_c only_c64 only_split; do
  echo "== class-width layout $mode ($(date +%T))"
  GS_SCALING=1 timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_$mode reddit-syn 128 $mode > $O/layout_$mode.log 2>&1 || echo "FAILED $mode"
done
echo "== class-layer kernels, memory side ($(date +%T))"
PMC_PROG="tools/bench_class.py 232965 10 noabl" PMC_MATCH="class_,gemm_,slab_reduce" timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_class_mem > $O/class_mem.log 2>&1 || echo "FAILED class mem"
echo "== done ($(date +%T))"
