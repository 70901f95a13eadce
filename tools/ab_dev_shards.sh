# device-resident G1 calls cut into record shards with the sort stage of shard s + 1 beside the accumulate of shard s (EIP2537_DEV_STAGES)
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/devsh; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
one() { python bench.py --workload g1msm --log2n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1 DEV_STAGES=${EIP2537_DEV_STAGES:-default} OVERLAP=${EIP2537_SORT_OVERLAP:-1}', 'shards', d['plan'].get('shards'), 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do
  EIP2537_DEV_STAGES=1 one 20; one 20; EIP2537_SORT_OVERLAP=0 one 20
  for st in 2 3 4 "1,3" "1,2,2" "1,4,4" "1,3,3,3" "1,2,2,2,2" "1,8,8" "1,4,4,4,4"; do EIP2537_DEV_STAGES=$st one 20; done
done | tee $O/ab20.txt
for l in 21 22; do EIP2537_DEV_STAGES=1 one $l; one $l; EIP2537_DEV_STAGES=4 one $l; EIP2537_DEV_STAGES=8 one $l; done | tee $O/ab_big.txt
