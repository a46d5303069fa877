#!/bin/bash
# round 6, batch 1 (one gpurun call): the GPU suite, then the GEMM skeleton pricing, then lane on/off A/B/A/B.
# A step that was killed at its limit (124 / 137) ends the batch: no GPU step is started after a hang.
O=gpurun_out/r6; mkdir -p $O
step() { name=$1; lim=$2; shift 2; echo "== $name"; timeout -k 10 $lim "$@" > $O/$name.log 2>&1; rc=$?; echo "   rc $rc"; tail -4 $O/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi; }
step b_tests 1000 python3 -m pytest tests -m gpu -q
step gemm_price 120 build/gemm_bf16x3 232965 602 20 price
step gemm_price_3rounds 120 build/gemm_bf16x3 196608 602 20 price
step gemm_price_4rounds 120 build/gemm_bf16x3 262144 602 20 price
