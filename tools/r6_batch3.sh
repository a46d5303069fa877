#!/bin/bash
# round 6, batch 3: the early dropout draw in the aggregation — parity tests that cover it, then the epoch's timers
O=gpurun_out/r6; mkdir -p $O
step() { name=$1; lim=$2; shift 2; echo "== $name"; timeout -k 10 $lim "$@" > $O/$name.log 2>&1; rc=$?; echo "   rc $rc"; tail -${TAILN:-4} $O/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi; }
step d_tests 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x
for rep in 1 2; do
  step bench_early_$rep 300 python3 bench.py --steps 300 --warmup 20 --bursts 1 --no-extras --no-cpu-baseline
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_early_$rep.log') if l.startswith('{')][-1])
print('epochs/s', round(d['value'],2), 'bursts', d['bursts']['epochs_per_s'], 'breakdown', d['breakdown_ms_per_epoch'], 'agg launch ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done
