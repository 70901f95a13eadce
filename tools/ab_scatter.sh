#!/bin/bash
# usage: tools/ab_scatter.sh (GPU box, repo root): scatter in 1 / 2 / 4 / 8 bucket-range passes at 2^20 and 2^22
R=$PWD
for lg in 20 22; do
for sp in 1 2 4 8 16; do
  EIP2537_SCATTER_PASSES=$sp timeout -k 10 100 python bench.py --log2n $lg --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$lg passes $sp', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
done
done
cd /tmp && export TMPDIR=/tmp
for sp in 1 4; do
EIP2537_SCATTER_PASSES=$sp timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2s/ks_sp$sp -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-host-abi > $R/gpurun_out/r2s/ks_sp$sp.log 2>&1
grep -h "k_msm_scatter" $R/gpurun_out/r2s/ks_sp$sp/*/*kernel_stats.csv | cut -c1-120
EIP2537_SCATTER_PASSES=$sp timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2s/wr_sp$sp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-host-abi > $R/gpurun_out/r2s/wr_sp$sp.log 2>&1
grep -h "k_msm_scatter" $R/gpurun_out/r2s/wr_sp$sp/*/*counter_collection.csv | awk -F, '{print $NF}' | sort -n | tail -1
done
