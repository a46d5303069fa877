#!/bin/bash
# Round-5 measurement batch (run on the GPU box through gpurun): writes everything under gpurun_out/r5m/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5m
mkdir -p $O
export GCN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
cd $R
step() { echo "== $1 ($(date +%T))"; }
step "kernel-trace stats of the timed region (bench.py --profile-run)"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 $R/bench.py --profile-run --steps 50 --warmup 5 > $O/profile_run.json 2> $O/profile_run.err) || echo "FAILED profile run"
step "PMC: dense first-layer products (bf16x3 default, f32 with GCNHIP_GEMM_BF16X3=0)"
PMC_MATCH=bf16x3 timeout -k 10 500 tools/pmc_gemm.sh $O/pmc_gemm_bx > $O/gemm_bf16x3_pmc.json 2> $O/gemm_bf16x3_pmc.err || echo "FAILED pmc gemm bx"
GCNHIP_GEMM_BF16X3=0 PMC_MATCH=t128,persist timeout -k 10 500 tools/pmc_gemm.sh $O/pmc_gemm_f32 > $O/gemm_f32_pmc.json 2> $O/gemm_f32_pmc.err || echo "FAILED pmc gemm f32"
step "PMC: class-layer kernels"
PMC_PROG="tools/bench_ops.py reddit-syn 128 small" PMC_MATCH="rowstream,atb,xent,slab_reduce" timeout -k 10 600 tools/pmc_gemm.sh $O/pmc_class > $O/class_layer_pmc.json 2> $O/class_layer_pmc.err || echo "FAILED pmc class"
step "PMC: hidden-width aggregation (factored), reddit-syn"
GS_SCALING=2 timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_gs reddit-syn 128 only_h > $O/graphsum_pmc.log 2>&1 || echo "FAILED pmc graphsum"
step "class-width layouts"
GS_SCALING=1 timeout -k 10 300 python tools/bench_ops.py reddit-syn 128 graphsum split41 > $O/class_layouts.log 2>&1 || echo "FAILED layouts"
step "structure experiments"
for d in reddit-syn-h03 reddit-syn-zipf reddit-syn; do timeout -k 10 300 python tools/exp_structure.py $d $O/exp_structure_$d.json > $O/exp_structure_$d.log 2>&1 || echo "FAILED exp $d"; done
step "done"
