export GPU_MAX_HW_QUEUES=16
timeout -k 10 700 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_long.py --seconds 90 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 40 --threads 4 2>&1 | tail -1
EIP2537_HIP_COALESCE=0 timeout -k 10 200 python tools/fuzz_long.py --seconds 45 --threads 4 --seed 9 2>&1 | tail -1
one() { python bench.py --workload g2msm --log2n $1 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('g2msm 2^$1 RCP8=${EIP2537_REDUCE_RCP8:-1}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 7 10 11 14 16 18; do EIP2537_REDUCE_RCP8=0 one $l; one $l; done; done | tee gpurun_out/g2_vec_horner.txt
