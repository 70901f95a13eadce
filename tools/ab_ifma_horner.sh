# host tail of a multiexp: doubling chains on AVX-512 IFMA vectors (ifma_horner.h) against the scalar code (EIP2537_HOST_IFMA=0); G1 opt-in
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/ifmah; mkdir -p $O
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc tools/ifma_check.hip -o /tmp/ifma_check 2>/dev/null && /tmp/ifma_check | tee $O/ifma_check.txt
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q failed $O/pytest.log && exit 1
timeout -k 10 200 python tools/fuzz_long.py --seconds 60 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 40 --threads 4 2>&1 | tail -1
one() { python bench.py --workload $1 --log2n $2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2 HOST_IFMA=${EIP2537_HOST_IFMA:-1} G1=${EIP2537_HOST_IFMA_G1:-0}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do
  for l in 7 10 16 18; do EIP2537_HOST_IFMA=0 one g2msm $l; one g2msm $l; done
  for l in 7 16 20; do EIP2537_HOST_IFMA=0 one g1msm $l; one g1msm $l; EIP2537_HOST_IFMA_G1=1 one g1msm $l; done
done | tee $O/ab.txt
