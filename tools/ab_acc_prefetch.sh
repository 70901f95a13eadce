export GPU_MAX_HW_QUEUES=16
one() { python bench.py --workload g1msm --log2n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1 ${EIP2537_HIP_LIB##*/}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], r['kernel'], '%.3f'%r['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2 3; do for l in 20 18 22; do one $l; EIP2537_HIP_LIB=$PWD/variants/libeip2537_hip_accpf.so one $l; done; done
