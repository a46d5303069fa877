#!/bin/bash
# round 6, verdict item 6: the validation lane on / off, A/B/A/B on ONE box, 1 500 timed epochs each, headline graph and the
# structure-free one.  One JSON line per run in $O/lane_ab.jsonl (value = epochs/s of the 1 500-epoch region).
O=gpurun_out/r6; mkdir -p $O; : > $O/lane_ab.jsonl
for ds in reddit-syn reddit-syn-h0; do
  for rep in 1 2; do
    for lane in on off; do
      timeout -k 10 200 python3 bench.py --dataset $ds --steps 1500 --warmup 20 --bursts 0 --no-extras --no-cpu-baseline --eval-lane $lane 2>> $O/lane_ab.err | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'dataset':'$ds','lane':'$lane','rep':$rep,'epochs_per_s':d['value'],'ms':d['ms_per_step'],'schedule':d['config']['aggregation_schedule'],'slice':d['config']['aggregation_slice_floats']}))" >> $O/lane_ab.jsonl || { echo "run failed ($ds $lane)"; exit 1; }
      tail -1 $O/lane_ab.jsonl
    done
  done
done
