for wl in "g1msm 20" "g1msm 17" "g1msm 16" "g1msm 12" "g1msm 7" "g2msm 18" "g2msm 16" "g2msm 10" "g2msm 7" "pairing 12" "pairing 6" "pairing 3"; do
  set -- $wl
  timeout -k 10 200 python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %s %.3f'%(d['roofline']['kernel'], d['roofline']['kernel_ms']), 'exact', d['bit_exact_vs_golden'])"
done
