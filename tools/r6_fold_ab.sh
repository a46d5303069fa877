#!/bin/bash
# round 6, verdict item 6 (second half): with the validation lane on, the training stream's segment sum inside the aggregation
# launch (gs_fold, experiments build) against the finalize launch that queues behind the lane's GEMM.  A/B/A/B, 1 500 epochs each.
O=gpurun_out/r6; mkdir -p $O; : > $O/fold_ab.jsonl
python3 -c "from cuda_gcn_amd import _lib; print('experiments build:', _lib.gcnhip().gcnhip_experiments())"
for rep in 1 2; do
  for mode in default fold_training fold_both; do
    case $mode in default) E="";; fold_training) E="HIPGCN_FOLD_TRAINING=1";; fold_both) E="GCNHIP_GS_FOLD=1";; esac
    env $E timeout -k 10 200 python3 bench.py --steps 1500 --warmup 20 --bursts 0 --no-extras --no-cpu-baseline --eval-lane on 2>> $O/fold_ab.err | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'mode':'$mode','rep':$rep,'epochs_per_s':d['value'],'ms':d['ms_per_step'],'final':d['final']}))" >> $O/fold_ab.jsonl || { echo "run failed ($mode)"; exit 1; }
    tail -1 $O/fold_ab.jsonl
  done
done
