#!/bin/bash
# usage: tools/ab_groups.sh   (GPU box, repo root): G1 MSM 2^20 / 2^18 / 2^22 with 1, 2, 4, 8 window groups
for lg in 20 18 22; do
  for g in 1 2 4 8; do
    EIP2537_MSM_GROUPS=$g python bench.py --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('log2n $lg groups $g', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum span %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
  done
done
