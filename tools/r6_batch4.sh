#!/bin/bash
# round 6, batch 4: MFMA shapes on random data; BASELINE configs[4] at its own width under partitioning (slow test, once)
O=gpurun_out/r6; mkdir -p $O
step() { name=$1; lim=$2; shift 2; echo "== $name"; timeout -k 10 $lim "$@" > $O/$name.log 2>&1; rc=$?; echo "   rc $rc"; tail -${TAILN:-4} $O/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi; }
TAILN=14 step mfma_patterns2 60 build/mfma_bf16_patterns
export GCN_RUN_SLOW=1
TAILN=6 step rmat21_256_p8 500 python3 -m pytest "tests/test_multirank.py::test_full_size_eight_logical_ranks_match_single_gpu" -m gpu -q -k "rmat-21-256" --durations=3
TAILN=6 step rmat22_256_p8 900 python3 -m pytest "tests/test_multirank.py::test_full_size_eight_logical_ranks_match_single_gpu" -m gpu -q -k "rmat-22-256" --durations=3
