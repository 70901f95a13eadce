#!/bin/bash
# usage: tools/ab_reduce.sh  (GPU box, repo root): reduce kernels with / without the one-wave-per-SIMD claim, block targets
for L in lib_excl0.so lib_excl1.so; do
  for rb in 100 140 200 256; do
    for wl in "g1msm 20" "g1msm 22"; do
      set -- $wl
      EIP2537_REDUCE_BLOCKS=$rb EIP2537_HIP_LIB=$PWD/variants/$L python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L rb $rb $1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
    done
  done
  for wl in "g1msm 16" "g1msm 12" "g2msm 16" "g2msm 10"; do
      set -- $wl
      EIP2537_HIP_LIB=$PWD/variants/$L python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L $1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
  done
done
