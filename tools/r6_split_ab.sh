#!/bin/bash
# round 6: rows cut into segments above SPLIT_EDGES edges (default 1024) need a finalize launch after every aggregation (5 per epoch).
# With no row cut at all (split_edges above the largest degree) the launch disappears and hub rows run on one wave each. A/B/A/B.
O=gpurun_out/r6; mkdir -p $O; touch $O/split_ab.jsonl
for rep in 1 2; do
  for se in ${SPLITS:-1024 4096 16384}; do
    env GCNHIP_SPLIT_EDGES=$se timeout -k 10 200 python3 bench.py --steps 1000 --warmup 20 --bursts 0 --no-extras --no-cpu-baseline 2>> $O/split_ab.err | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'split_edges':$se,'rep':$rep,'epochs_per_s':d['value'],'ms':d['ms_per_step'],'agg_ms':d['roofline']['avg_launch_ms'],'breakdown':d['breakdown_ms_per_epoch'],'final':d['final']}))" >> $O/split_ab.jsonl || { echo "run failed ($se)"; tail -5 $O/split_ab.err; exit 1; }
    tail -1 $O/split_ab.jsonl | cut -c1-330
  done
done
