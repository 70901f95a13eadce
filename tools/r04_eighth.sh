#!/bin/bash
# round 4, eighth GPU call: G1 window sums on the device (c <= 13 plans): correctness + sizes; the in-library split over 4 pools of one GPU
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04h; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
one() { python bench.py --workload $1 --log2n $2 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'dominant %.3f'%r['kernel_ms'], 'sort %.3f reduce %.3f'%(r.get('sort_stage_ms',0), r.get('fold_reduce_ms',0)), 'exact', d['bit_exact_vs_golden'])"; }
for c in "g1msm 17" "g1msm 16" "g1msm 14" "g1msm 12" "g1msm 10" "g1msm 7" "g2msm 16" "g2msm 10"; do one $c >> $O/sizes.txt; done
cat $O/sizes.txt
for p in default 0; do
  echo "EIP2537_HIP_DEVICES=0,0,0,0 EIP2537_H2D_PIPELINE=$p" >> $O/split4.txt
  if [ $p = default ]; then EIP2537_HIP_DEVICES=0,0,0,0 timeout -k 10 300 python tools/dbg_host_abi.py 22 g1 >> $O/split4.txt 2>&1
  else EIP2537_H2D_PIPELINE=0 EIP2537_HIP_DEVICES=0,0,0,0 timeout -k 10 300 python tools/dbg_host_abi.py 22 g1 >> $O/split4.txt 2>&1; fi
done
EIP2537_HIP_DEVICES=0,0 timeout -k 10 300 python tools/dbg_host_abi.py 21 g1 >> $O/split4.txt 2>&1
grep -v amdgpu.ids $O/split4.txt
