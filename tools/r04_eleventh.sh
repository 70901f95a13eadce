#!/bin/bash
# round 4, eleventh GPU call: window-width re-sweep around the plan boundaries with the round-4 kernels; RCCL single-rank line
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04k; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
one() { python bench.py --workload $1 --log2n $2 --window $3 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2 c=$3', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'dominant %s %.3f'%(r['kernel'], r['kernel_ms']), 'sort %.3f reduce %.3f'%(r.get('sort_stage_ms',0), r.get('fold_reduce_ms',0)), 'exact', d['bit_exact_vs_golden'])"; }
for c in 12 13 14 15 16; do one g1msm 16 $c; done | tee -a $O/window_sweep.txt
for c in 13 14 15 16; do one g1msm 17 $c; done | tee -a $O/window_sweep.txt
for c in 13 15 16; do one g1msm 18 $c; done | tee -a $O/window_sweep.txt
for c in 12 13 14 15 16; do one g2msm 16 $c; done | tee -a $O/window_sweep.txt
for c in 11 12 13; do one g2msm 14 $c; done | tee -a $O/window_sweep.txt
for c in 10 11 12 13; do one g1msm 13 $c; done | tee -a $O/window_sweep.txt
BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err; echo "rccl rc=$?"
python -c "
import json;d=json.load(open('$O/bench_1rank_rccl.json')); print(d['ms_per_step'], d['step_ms'], d['config']['workload'], d['bit_exact_vs_golden'])"
