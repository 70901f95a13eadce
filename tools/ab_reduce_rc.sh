export GPU_MAX_HW_QUEUES=16
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
one() { python bench.py --workload g1msm --log2n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0), 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do
for lib in variants/libeip2537_hip_prerc.so blst_eip2537_amd/libeip2537_hip.so; do
  echo "# $lib"; EIP2537_HIP_LIB=$PWD/$lib one 20; EIP2537_HIP_LIB=$PWD/$lib one 18
  EIP2537_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 2>&1 | grep -v amdgpu.ids
done; done
