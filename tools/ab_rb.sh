#!/bin/bash
# usage: tools/ab_rb.sh (GPU box, repo root): reduce block target sweep at 2^20, then a size sweep
for rb in 136 200 240 250 256; do
  EIP2537_REDUCE_BLOCKS=$rb python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rb $rb', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
done
for wl in "g1msm 22" "g1msm 18" "g1msm 16" "g1msm 12" "g1msm 7" "g2msm 16" "g2msm 10"; do
      set -- $wl
      python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
done
