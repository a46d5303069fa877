#!/bin/bash
# Round-5 measurement batch, part 3 (after the chunk-major keep words): the driver-style bench line, the kernel-trace stats of
# the timed region, the counters of the first-layer products, the command line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5k
mkdir -p $O
export GCN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
cd $R
step() { echo "== $1 ($(date +%T))"; }
step "bench.py as the driver runs it"
timeout -k 10 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || echo "FAILED bench"
step "kernel-trace stats of the timed region (bench.py --profile-run)"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 $R/bench.py --profile-run --steps 50 --warmup 5 > $O/profile_run.json 2> $O/profile_run.err) || echo "FAILED profile run"
python tools/kernel_stats.py $O/prof_run > $O/bench_n1_kernels.txt 2>&1 || echo "FAILED kernel_stats"
step "PMC: dense first-layer products (bf16x3)"
PMC_MATCH=bf16x3 timeout -k 10 500 tools/pmc_gemm.sh $O/pmc_gemm_bx > $O/gemm_bf16x3_pmc.json 2> $O/gemm_bf16x3_pmc.err || echo "FAILED pmc gemm bx"
step "the command line on reddit-syn"
timeout -k 10 400 python tools/run_cli_reddit.py --out $O/cli_reddit.json > $O/cli_reddit.log 2>&1 || echo "FAILED cli"
step "done"
