#!/bin/bash
# device pipeline time of an MSM as a function of the window width c (planner check)
for l in ${LOGS:-12 14 16}; do
  for c in ${CS:-0 8 9 10 11 12 13 14 15 16}; do
    python bench.py --workload ${WL:-g1msm} --log2n $l --window $c --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/ws_tmp.json 2>/dev/null || { echo "2^$l c=$c FAILED"; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/ws_tmp.json')); print('2^$l c=$c', 'ms/step %.3f'%d['ms_per_step'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'])"
  done
done
