export GPU_MAX_HW_QUEUES=16
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
one() { python bench.py --workload g1msm --log2n $1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1 DEV_STAGES=${EIP2537_DEV_STAGES:-default}', 'shards', d['plan'].get('shards'), 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for l in 20 21 22; do EIP2537_DEV_STAGES=1 one $l; one $l; done | tee gpurun_out/devsh/final.txt
