#!/bin/bash
# usage: ab.sh lib1 lib2 ...   (run from repo root on the GPU box)
for L in "$@"; do
  echo "== $L" >> gpurun_out/ab.txt
  for wl in ${WLS:-pairing g2msm g1msm}; do
    EIP2537_HIP_LIB=$PWD/variants/$L timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || { echo "$wl FAILED" >> gpurun_out/ab.txt; tail -3 gpurun_out/ab_tmp.err >> gpurun_out/ab.txt; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl', 'ms/step %.3f'%d['ms_per_step'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])" >> gpurun_out/ab.txt
  done
done
cat gpurun_out/ab.txt
