#!/bin/bash
# round 4, tenth GPU call: coarse scatter without its recount pass, tree form of the 4-lane small fold
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04j; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
one() { python bench.py --workload $1 --log2n $2 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'dominant %.3f'%r['kernel_ms'], 'sort %.3f reduce %.3f'%(r.get('sort_stage_ms',0), r.get('fold_reduce_ms',0)), 'exact', d['bit_exact_vs_golden'])"; }
for c in "g1msm 20" "g1msm 18" "g1msm 17" "g1msm 16" "g1msm 14" "g1msm 12" "g1msm 7" "g2msm 16"; do one $c >> $O/sizes.txt; done
cat $O/sizes.txt
timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 2>&1 | grep -v amdgpu.ids | tee -a $O/stages.txt
timeout -k 10 200 python tools/fuzz_long.py --seconds 60 --threads 4 --mid > $O/fuzz_mid.txt 2>&1; tail -1 $O/fuzz_mid.txt
timeout -k 10 200 python tools/fuzz_long.py --seconds 45 --threads 4 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
