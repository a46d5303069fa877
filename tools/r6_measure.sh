#!/bin/bash
# Round-6 measurement batch (one gpurun call, on the round's last tree): everything bench.py quotes from profiles/ taken again,
# each file carrying the sha256 of the kernel sources it describes (cuda_gcn_amd/provenance.py).  Writes under gpurun_out/r6m/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r6m
mkdir -p $O
export GCN_COMMIT=$(cat $R/.commit_for_profiles 2>/dev/null || echo unknown)
cd $R
step() { echo "== $1 ($(date +%T))"; }
step "kernel-trace stats of the timed region (bench.py --profile-run)"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_run -- python3 $R/bench.py --profile-run --steps 50 --warmup 5 > $O/profile_run.json 2> $O/profile_run.err) || echo "FAILED profile run"
python3 tools/kernel_stats.py $O/prof_run > $O/bench_n1_kernels.txt 2>&1 || echo "FAILED kernel_stats"
step "PMC: hidden-width aggregation (factored), reddit-syn"
GS_SCALING=2 PMC_SOURCES=graphsum.hip timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_gs reddit-syn 128 only_h > $O/graphsum_pmc.log 2>&1 || echo "FAILED pmc graphsum"
step "PMC: class-width aggregation"
GS_SCALING=1 PMC_SOURCES=graphsum.hip timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_gs_class reddit-syn 128 only_c > $O/graphsum_pmc_class.log 2>&1 || echo "FAILED pmc graphsum class"
step "PMC: hidden-width aggregation, R-MAT scale 21 (HBM regime)"
GS_SCALING=1 PMC_SOURCES=graphsum.hip timeout -k 10 600 tools/pmc_graphsum.sh $O/pmc_gs_rmat rmat-21 128 only_h > $O/graphsum_pmc_rmat.log 2>&1 || echo "FAILED pmc graphsum rmat"
step "measured ceiling of the cache-regime gather"
timeout -k 10 600 python3 tools/gather_peak.py --out $O/gather_peak.json > $O/gather_peak.log 2>&1 || echo "FAILED gather_peak"
step "PMC: dense first-layer products (bf16x3)"
PMC_MATCH=bf16x3 timeout -k 10 500 tools/pmc_gemm.sh $O/pmc_gemm_bx > $O/gemm_bf16x3_pmc.json 2> $O/gemm_bf16x3_pmc.err || echo "FAILED pmc gemm bx"
step "two-stream epoch, kernel by kernel"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/timeline -o ks -- python3 $R/bench.py --steps 10 --warmup 3 --bursts 0 --no-cpu-baseline --no-extras > $O/timeline.json 2> $O/timeline.err) || echo "FAILED timeline"
python3 tools/epoch_timeline.py $(find $O/timeline -name "ks_kernel_trace.csv" | head -1) > $O/two_stream_epoch_timeline.txt 2>&1 || echo "FAILED epoch_timeline"
step "the command line on reddit-syn"
timeout -k 10 400 python3 tools/run_cli_reddit.py --out $O/cli_reddit.json > $O/cli_reddit.log 2>&1 || echo "FAILED cli"
step "rmat-22 as a whole model"
timeout -k 10 500 python3 bench.py --dataset rmat-22 --steps 10 --warmup 2 --bursts 1 --no-cpu-baseline --no-extras > $O/bench_rmat22.json 2> $O/bench_rmat22.err || echo "FAILED rmat22"
step "done"
