#!/bin/bash
# round 4, ninth GPU call: pinned return buffer: correctness + sizes; staged-path fuzz; mid-size fuzz through the c = 16 plan
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04i; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
one() { python bench.py --workload $1 --log2n $2 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'exact', d['bit_exact_vs_golden'])"; }
for c in "g1msm 20" "g1msm 16" "g1msm 12" "g1msm 7" "g2msm 16" "g2msm 7" "pairing 12" "pairing 6"; do one $c >> $O/sizes.txt; done
cat $O/sizes.txt
EIP2537_H2D_STAGES=1,2,3,1 timeout -k 10 400 python tools/fuzz_staged.py --cases 14 --seed 1 > $O/fuzz_staged_a.txt 2>&1; tail -4 $O/fuzz_staged_a.txt
EIP2537_H2D_STAGES=3,1,1,1,1,2 timeout -k 10 400 python tools/fuzz_staged.py --cases 10 --seed 2 > $O/fuzz_staged_b.txt 2>&1; tail -3 $O/fuzz_staged_b.txt
timeout -k 10 200 python tools/fuzz_long.py --seconds 90 --threads 4 --mid --window 16 > $O/fuzz_mid16.txt 2>&1; tail -2 $O/fuzz_mid16.txt
timeout -k 10 200 python tools/fuzz_long.py --seconds 60 --threads 4 --mid > $O/fuzz_mid.txt 2>&1; tail -2 $O/fuzz_mid.txt
