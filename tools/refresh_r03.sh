#!/bin/bash
# Refresh the round-3 measurement artefacts (GPU box, repo root).  Part 1: per-kernel profiles with PMC passes, the traffic json
# bench.py quotes, the three bench lines.  Part 2 (tools/refresh_r03.sh part2): size sweep, small calls, concurrent callers, the
# 2-rank rehearsal, the long fuzz.  Results land in gpurun_out/final/; copy into profiles/ to commit.
O=gpurun_out/final; mkdir -p $O
step() { echo "== $* ($(date +%T))" | tee -a $O/progress.log; }
if [ "$1" != "part2" ]; then
step prof g1msm 2^20;  timeout -k 10 500 bash tools/prof_kernels.sh r03_g1msm_2p20 --steps 5 --warmup 2 --no-secondary > $O/prof_g1_20.log 2>&1 || exit 1
step prof pairing;     timeout -k 10 400 bash tools/prof_kernels.sh r03_pairing_2p12 --workload pairing --steps 10 --warmup 2 > $O/prof_pair.log 2>&1 || exit 1
step prof g2msm 2^16;  timeout -k 10 400 bash tools/prof_kernels.sh r03_g2msm_2p16 --workload g2msm --steps 10 --warmup 2 > $O/prof_g2.log 2>&1 || exit 1
cp gpurun_out/r03_*_kernels.csv $O/ && cp gpurun_out/r03_*_kernels.csv profiles/
python tools/make_traffic_json.py r03 g1msm:20:profiles/r03_g1msm_2p20_kernels.csv:k_msm_accum_l pairing:12:profiles/r03_pairing_2p12_kernels.csv:k_pair_lines8 \
       g2msm:16:profiles/r03_g2msm_2p16_kernels.csv:k_msm_accum2c > $O/traffic.log 2>&1 && cp profiles/r03_traffic.json $O/
step bench default;    timeout -k 10 600 python bench.py > $O/r03_bench_default.json 2> $O/bench_default.err || exit 1
step bench g2msm;      timeout -k 10 400 python bench.py --workload g2msm > $O/r03_bench_g2msm_2p16.json 2> $O/bench_g2.err || exit 1
step bench pairing;    timeout -k 10 400 python bench.py --workload pairing > $O/r03_bench_pairing_2p12.json 2> $O/bench_pair.err || exit 1
step part 1 done
else
step size sweep
rm -f $O/r03_size_sweep.txt
for wl in "g1msm 22" "g1msm 21" "g1msm 20" "g1msm 19" "g1msm 18" "g1msm 17" "g1msm 16" "g1msm 14" "g1msm 12" "g1msm 10" "g1msm 7" "g2msm 18" "g2msm 16" "g2msm 14" "g2msm 10" "g2msm 7" "pairing 12" "pairing 10" "pairing 6" "pairing 3"; do
  set -- $wl
  timeout -k 10 200 python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %s %.3f'%(d['roofline']['kernel'], d['roofline']['kernel_ms']), 'exact', d['bit_exact_vs_golden'])" | tee -a $O/r03_size_sweep.txt
done
step small calls;      timeout -k 10 500 python tools/small_calls.py > $O/r03_small_calls.txt 2>&1 || exit 1
step concurrency;      timeout -k 10 400 python tools/concurrency_timing.py --threads 1 16 64 > $O/r03_concurrent_callers.txt 2>&1 || exit 1
step 2 ranks over gloo; BENCH_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/r03_bench_2rank_gloo_rehearsal.json 2> $O/bench_2rank.err || exit 1
step fuzz;             timeout -k 10 300 python tools/fuzz_long.py --seconds 120 --threads 4 > $O/r03_fuzz_long.txt 2>&1 || exit 1
step part 2 done
fi
