#!/bin/bash
# round 6 experiment: the validation lane on n compute units of its own (HIPGCN_LANE_CUS), the training pass on the rest. A/B/A/B.
O=gpurun_out/r6; mkdir -p $O; : > $O/lane_cus_ab.jsonl
for rep in 1 2; do
  for n in 0 16 32 64; do
    E=""; [ $n -gt 0 ] && E="HIPGCN_LANE_CUS=$n"
    env $E timeout -k 10 200 python3 bench.py --steps 1000 --warmup 20 --bursts 0 --no-extras --no-cpu-baseline --eval-lane on 2>> $O/lane_cus_ab.err | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lane_cus':$n,'rep':$rep,'epochs_per_s':d['value'],'ms':d['ms_per_step'],'final':d['final']}))" >> $O/lane_cus_ab.jsonl || { echo "run failed (lane_cus $n)"; tail -5 $O/lane_cus_ab.err; exit 1; }
    tail -1 $O/lane_cus_ab.jsonl
  done
done
