# A/B of the limb-form G2 accumulate (k_msm_accum2c_l) against k_msm_accum2c (EIP2537_G2_LIMB=0); extra libraries given as arguments
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/g2l; mkdir -p $O
one() { python bench.py --workload g2msm --log2n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('g2 2^$1 LIMB=${EIP2537_G2_LIMB:-1} ${EIP2537_HIP_LIB##*/}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], r['kernel'], '%.3f'%r['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 10 14 16 18; do
  EIP2537_G2_LIMB=0 one $l; one $l
  for lib in "$@"; do EIP2537_HIP_LIB=$PWD/$lib one $l; done
done; done | tee $O/ab.txt
