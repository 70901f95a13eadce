#!/bin/bash
# round 4, fourth GPU call: per-kernel times of the 2^20 device-resident step (what did the fused sort stage cost?), slice A/B,
# pairing sweep beyond 2^12, degenerate inputs, 4-rank gloo rehearsal
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04d; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
bash tools/kstats.sh r04d_g1 bench.py --workload g1msm --steps 6 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 > $O/kstats_g1_2p20.txt 2>&1
cat $O/kstats_g1_2p20.txt
for sl in 32768 8192; do
  echo "EIP2537_SORT_SLICE=$sl" >> $O/slice_ab.txt
  EIP2537_SORT_SLICE=$sl timeout -k 10 200 python bench.py --workload g1msm --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('g1msm 2^20', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'sort %.3f reduce %.3f'%(d['roofline']['sort_stage_ms'], d['roofline']['fold_reduce_ms']))" >> $O/slice_ab.txt
done
cat $O/slice_ab.txt
for l in 13 14 16; do
  timeout -k 10 300 python bench.py --workload pairing --log2n $l --steps 6 --warmup 2 --no-cpu-baseline --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); v=d['roofline_valu']; print('pairing 2^$l', 'ms/check %.3f'%d['ms_per_step'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'pipeline frac of mad roof %.3f'%v['device_pipeline']['frac'], ' '.join('%s %.3f ms (%.2f)'%(k['kernel'],k['ms'],k['frac']) for k in v['kernels']), 'kernel', d['plan']['kernel'], 'one' , d['bit_exact_vs_golden'])" >> $O/pairing_sweep.txt
done
cat $O/pairing_sweep.txt
timeout -k 10 300 python tools/degenerate_timing.py > $O/degenerate.txt 2>&1; grep -v amdgpu.ids $O/degenerate.txt
BENCH_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 3 --warmup 1 > $O/bench_4rank_gloo.json 2> $O/bench_4rank_gloo.err; echo "4-rank rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench_4rank_gloo.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus','bit_exact_vs_golden')}); print(d.get('strong')); print(d.get('in_library_split'))"
