#!/bin/bash
# Refresh the round-2 measurement artefacts (GPU box, repo root): per-kernel profiles with PMC passes, the
# traffic json bench.py quotes, the three bench lines, the size sweep.  Results land in gpurun_out/final/.
O=gpurun_out/final; mkdir -p $O
step() { echo "== $* ($(date +%T))"; }
step prof g1msm 2^20;  timeout -k 10 500 bash tools/prof_kernels.sh r02_g1msm_2p20 --steps 5 --warmup 2 --no-secondary > $O/prof_g1_20.log 2>&1
python tools/traffic_json.py gpurun_out/r02_g1msm_2p20_kernels.csv g1msm 20 profiles/r02_traffic.json > /dev/null && cp profiles/r02_traffic.json $O/
step prof pairing;     timeout -k 10 400 bash tools/prof_kernels.sh r02_pairing_2p12 --workload pairing --steps 10 --warmup 2 > $O/prof_pair.log 2>&1
step prof g2msm 2^16;  timeout -k 10 400 bash tools/prof_kernels.sh r02_g2msm_2p16 --workload g2msm --steps 10 --warmup 2 > $O/prof_g2.log 2>&1
step prof g1msm 2^16;  timeout -k 10 400 bash tools/prof_kernels.sh r02_g1msm_2p16 --log2n 16 --steps 10 --warmup 2 --no-secondary > $O/prof_g1_16.log 2>&1
cp gpurun_out/r02_*_kernels.csv $O/
step bench default;    timeout -k 10 600 python bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
step bench g2msm;      timeout -k 10 400 python bench.py --workload g2msm > $O/r02_bench_g2msm_2p16.json 2> $O/bench_g2.err
step bench pairing;    timeout -k 10 400 python bench.py --workload pairing > $O/r02_bench_pairing_2p12.json 2> $O/bench_pair.err
step size sweep
for wl in "g1msm 22" "g1msm 21" "g1msm 20" "g1msm 19" "g1msm 18" "g1msm 17" "g1msm 16" "g1msm 14" "g1msm 12" "g1msm 10" "g1msm 7" "g2msm 18" "g2msm 16" "g2msm 14" "g2msm 10" "g2msm 7" "pairing 12" "pairing 10" "pairing 6" "pairing 3"; do
  set -- $wl
  timeout -k 10 200 python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %s %.3f'%(d['roofline']['kernel'], d['roofline']['kernel_ms']), 'exact', d['bit_exact_vs_golden'])" | tee -a $O/r02_size_sweep.txt
done
step done
