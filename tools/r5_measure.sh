#!/bin/bash
# Round-5 measurement batch (run on the GPU box through gpurun): writes everything under gpurun_out/r5f/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5f
mkdir -p $O
export GCN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
cd $R
step() { echo "== $1 ($(date +%T))"; }
step "kernel-trace stats of the timed region (bench.py --profile-run)"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 $R/bench.py --profile-run --steps 50 --warmup 5 > $O/profile_run.json 2> $O/profile_run.err) || echo "FAILED profile run"
step "PMC: dense first-layer products (bf16x3 default, f32 with GCNHIP_GEMM_BF16X3=0)"
PMC_MATCH=bf16x3 timeout -k 10 500 tools/pmc_gemm.sh $O/pmc_gemm_bx > $O/gemm_bf16x3_pmc.json 2> $O/gemm_bf16x3_pmc.err || echo "FAILED pmc gemm bx"
step "PMC: class-layer kernels (bf16x3 default; f32 row-stream kernels with GCNHIP_GEMM_BF16X3=0)"
PMC_PROG="tools/bench_class.py 232965 10 noabl" PMC_MATCH="class_,slab_reduce,rowstream,atb" timeout -k 10 600 tools/pmc_gemm.sh $O/pmc_class > $O/class_layer_pmc.json 2> $O/class_layer_pmc.err || echo "FAILED pmc class"
step "PMC: hidden-width aggregation (factored), reddit-syn"
GS_SCALING=2 timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_gs reddit-syn 128 only_h > $O/graphsum_pmc.log 2>&1 || echo "FAILED pmc graphsum"
step "class layer: timings and ablations"
timeout -k 10 200 python tools/bench_class.py > $O/class_layer_bench.log 2>&1 || echo "FAILED bench_class"
step "the command line on reddit-syn"
timeout -k 10 400 python tools/run_cli_reddit.py --out $O/cli_reddit.json > $O/cli_reddit.log 2>&1 || echo "FAILED cli"
step "rmat-22 as a whole model"
timeout -k 10 500 python bench.py --dataset rmat-22 --steps 10 --warmup 2 --bursts 1 --no-cpu-baseline --no-extras > $O/bench_rmat22.json 2> $O/bench_rmat22.err || echo "FAILED rmat22"
step "done"
