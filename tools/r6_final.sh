#!/bin/bash
# round 6, last batch: the GPU suite on the default build at the last tree, then bench.py as the driver runs it
O=gpurun_out/r6f; mkdir -p $O
echo "== tests"; timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $O/final_tests.log 2>&1; rc=$?; tail -3 $O/final_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at their limit: stopping"; exit 1; fi
echo "== smoke"; timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench"; timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -13 $O/bench_n1.err
