#!/bin/bash
# round 6 experiment: the validation lane released at the start of the training pass with its GEMM in the four-wave form. A/B/A/B.
O=gpurun_out/r6; mkdir -p $O; : > $O/lane_early_ab.jsonl
for rep in 1 2; do
  for mode in default early4 early8 late4; do
    case $mode in default) E="";; early4) E="HIPGCN_LANE_EARLY=1 GCNHIP_GEMM_LANE_WAVES=4";; early8) E="HIPGCN_LANE_EARLY=1";; late4) E="GCNHIP_GEMM_LANE_WAVES=4";; esac
    env $E timeout -k 10 200 python3 bench.py --steps 1000 --warmup 20 --bursts 0 --no-extras --no-cpu-baseline --eval-lane on 2>> $O/lane_early_ab.err | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'mode':'$mode','rep':$rep,'epochs_per_s':d['value'],'ms':d['ms_per_step'],'final':d['final']}))" >> $O/lane_early_ab.jsonl || { echo "run failed ($mode)"; tail -5 $O/lane_early_ab.err; exit 1; }
    tail -1 $O/lane_early_ab.jsonl
  done
done
