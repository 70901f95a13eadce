export GPU_MAX_HW_QUEUES=16
R=$PWD
bash tools/kstats.sh g2_16 $R/bench.py --workload g2msm --log2n 16 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 | grep -v "^{" 
bash tools/kstats.sh g1_16 $R/bench.py --workload g1msm --log2n 16 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 | grep -v "^{"
one() { python bench.py --workload g2msm --log2n $1 --window $2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('g2 2^$1 c=$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], r['kernel'], '%.3f'%r['kernel_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0))"; }
for cfg in "12 8" "12 11" "13 11" "13 13" "14 11" "14 13" "16 11" "16 13" "17 13" "17 16" "18 13" "18 16" "19 13" "19 16"; do one $cfg; done | tee gpurun_out/g2_window_sweep.txt
