export GPU_MAX_HW_QUEUES=16
one() { python bench.py --workload $1 --log2n $2 --window $3 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2 c=$3', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], r['kernel'], '%.3f'%r['kernel_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0))"; }
for rep in 1 2; do
for l in 11 12 13; do for c in 8 11 13; do one g1msm $l $c; done; done
for l in 11 12 13; do for c in 8 11 13; do one g2msm $l $c; done; done
done | tee gpurun_out/band_sweep.txt
