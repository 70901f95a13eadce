#!/bin/bash
# round 4, fifth GPU call: sort stage with the fused steps reverted: correctness, per-kernel times, slice A/B, stage sweep, 2^16 sizes
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04e; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
bash tools/kstats.sh r04e_g1 $R/bench.py --workload g1msm --steps 6 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 > $O/kstats_g1_2p20.txt 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $O/kstats_g1_2p20.txt
one() { python bench.py --workload $1 --log2n $2 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'dominant %.3f'%r['kernel_ms'], 'sort %.3f reduce %.3f'%(r.get('sort_stage_ms',0), r.get('fold_reduce_ms',0)), 'exact', d['bit_exact_vs_golden'])"; }
for sl in default 16384; do
  echo "EIP2537_SORT_SLICE=$sl" >> $O/slice_ab.txt
  if [ $sl = default ]; then one g1msm 20 >> $O/slice_ab.txt; else EIP2537_SORT_SLICE=$sl one g1msm 20 >> $O/slice_ab.txt; fi
done
one g1msm 18 >> $O/slice_ab.txt; one g1msm 16 >> $O/slice_ab.txt; one g2msm 16 >> $O/slice_ab.txt; one g1msm 12 >> $O/slice_ab.txt
cat $O/slice_ab.txt
for st in default "1,3,4,4,4" "1,4,4,4,3" "1,3,4,4,3,1" "1,4,4,4,2,1" "2,5,5,4"; do
  if [ "$st" = default ]; then timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  else EIP2537_H2D_STAGES=$st timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1; fi
done
EIP2537_SORT_SLICE=32768 EIP2537_H2D_STAGES=1,3,4,4,4 timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
grep -v amdgpu.ids $O/stages.txt
