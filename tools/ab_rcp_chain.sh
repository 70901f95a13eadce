# row / column sums of the c = 13 two-level reduce: one lane per chain of 4 buckets (EIP2537_RCP_CHAIN=4) against 4-lane groups per chain of 8 / 16
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/rcp4; mkdir -p $O
for ch in 8 16; do EIP2537_RCP_CHAIN=$ch timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1; done
EIP2537_RCP_CHAIN=8 timeout -k 10 100 python tools/fuzz_long.py --mid --seconds 30 --threads 4 2>&1 | tail -1
EIP2537_RCP_CHAIN=16 timeout -k 10 100 python tools/fuzz_long.py --window 13 --seconds 30 --threads 4 2>&1 | tail -1
one() { python bench.py --workload g1msm --log2n $1 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1 RCP_CHAIN=${EIP2537_RCP_CHAIN}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0), 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 14 16 17; do for ch in 4 8 16; do EIP2537_RCP_CHAIN=$ch one $l; done; done; done | tee $O/ab.txt
