#!/bin/bash
# round 6, batch 2: MFMA issue patterns, the launcher's hang test alone, lane on/off A/B/A/B
O=gpurun_out/r6; mkdir -p $O
step() { name=$1; lim=$2; shift 2; echo "== $name"; timeout -k 10 $lim "$@" > $O/$name.log 2>&1; rc=$?; echo "   rc $rc"; tail -${TAILN:-4} $O/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi; }
TAILN=8 step mfma_patterns 60 build/mfma_bf16_patterns
step c_bench_tests 700 python3 -m pytest tests/test_bench.py -m gpu -q
TAILN=10 step lane_ab 900 bash tools/r6_lane_ab.sh
