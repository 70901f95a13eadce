export GPU_MAX_HW_QUEUES=16
timeout -k 10 700 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 60 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --window 11 --seconds 30 --threads 4 2>&1 | tail -1
for wl in "g1msm 12" "g1msm 13" "g2msm 12" "g2msm 13"; do set -- $wl
python bench.py --workload $1 --log2n $2 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'c', d['plan']['window_bits'], 'ms/step %.3f'%d['ms_per_step'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], d['bit_exact_vs_golden'])"; done
