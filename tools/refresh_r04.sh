#!/bin/bash
# Refresh the round-4 measurement artefacts (GPU box, repo root).  Part 1: per-kernel profiles with PMC passes, the traffic json
# bench.py quotes, the three bench lines.  Part 2 (tools/refresh_r04.sh part2): size sweep, native concurrent callers against the
# round-2 library, in-library split on one GPU, degenerate inputs, the 4-rank rehearsal, the long fuzz.
# part3: the size sweep, host-ABI sizes and degenerate inputs of part 2 only.
# Results land in gpurun_out/final4/; copy into profiles/ to commit.
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/final4; mkdir -p $O
MODE=$1
step() { echo "== $* ($(date +%T))" | tee -a $O/progress.log; }
if [ "$MODE" != "part2" ] && [ "$MODE" != "part3" ]; then
step prof g1msm 2^20;  timeout -k 10 500 bash tools/prof_kernels.sh r04_g1msm_2p20 --steps 5 --warmup 2 --no-secondary > $O/prof_g1_20.log 2>&1 || exit 1
step prof pairing;     timeout -k 10 400 bash tools/prof_kernels.sh r04_pairing_2p12 --workload pairing --steps 10 --warmup 2 > $O/prof_pair.log 2>&1 || exit 1
step prof g2msm 2^16;  timeout -k 10 400 bash tools/prof_kernels.sh r04_g2msm_2p16 --workload g2msm --steps 10 --warmup 2 > $O/prof_g2.log 2>&1 || exit 1
step prof g1msm 2^16;  timeout -k 10 400 bash tools/prof_kernels.sh r04_g1msm_2p16 --workload g1msm --log2n 16 --steps 10 --warmup 2 --no-secondary > $O/prof_g1_16.log 2>&1 || exit 1
cp gpurun_out/r04_*_kernels.csv $O/ && cp gpurun_out/r04_*_kernels.csv profiles/
python tools/make_traffic_json.py r04 g1msm:20:profiles/r04_g1msm_2p20_kernels.csv:k_msm_accum_l pairing:12:profiles/r04_pairing_2p12_kernels.csv:k_pair_lines8 \
       g2msm:16:profiles/r04_g2msm_2p16_kernels.csv:k_msm_accum2c_l > $O/traffic.log 2>&1 && cp profiles/r04_traffic.json $O/
step bench default;    timeout -k 10 600 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err || exit 1
step bench g2msm;      timeout -k 10 400 python bench.py --workload g2msm > $O/r04_bench_g2msm_2p16.json 2> $O/bench_g2.err || exit 1
step bench pairing;    timeout -k 10 400 python bench.py --workload pairing > $O/r04_bench_pairing_2p12.json 2> $O/bench_pair.err || exit 1
step part 1 done
else
step size sweep
rm -f $O/r04_size_sweep.txt
for wl in "g1msm 22" "g1msm 21" "g1msm 20" "g1msm 19" "g1msm 18" "g1msm 17" "g1msm 16" "g1msm 14" "g1msm 12" "g1msm 10" "g1msm 7" "g2msm 18" "g2msm 16" "g2msm 14" "g2msm 10" "g2msm 7" "pairing 12" "pairing 10" "pairing 6" "pairing 3"; do
  set -- $wl
  timeout -k 10 200 python bench.py --workload $1 --log2n $2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 2^$2', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %s %.3f'%(d['roofline']['kernel'], d['roofline']['kernel_ms']), 'exact', d['bit_exact_vs_golden'])" | tee -a $O/r04_size_sweep.txt
done
step host-ABI sizes
for l in 18 19 20 21 22; do timeout -k 10 200 python tools/dbg_host_abi.py $l g1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_host_abi_sizes.txt; done
EIP2537_H2D_PIPELINE=0 timeout -k 10 200 python tools/dbg_host_abi.py 20 g1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_host_abi_sizes.txt
timeout -k 10 200 python tools/dbg_host_abi.py 18 g2 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_host_abi_sizes.txt
[ "$MODE" = "part3" ] && { step degenerate; timeout -k 10 300 python tools/degenerate_timing.py 2>&1 | grep -v amdgpu.ids > $O/r04_degenerate_inputs.txt; step part 3 done; exit 0; }
step in-library split, one GPU
for dv in "0,0" "0,0,0,0"; do
  echo "EIP2537_HIP_DEVICES=$dv" | tee -a $O/r04_in_library_split.txt
  EIP2537_HIP_DEVICES=$dv timeout -k 10 300 python tools/dbg_host_abi.py 22 g1 2>&1 | grep -v amdgpu.ids | tee -a $O/r04_in_library_split.txt
done
step native concurrency
gcc -O2 tools/conc_bench.c -ldl -lpthread -o /tmp/conc_bench
for lib in variants/libeip2537_hip_r2.so blst_eip2537_amd/libeip2537_hip.so; do
  [ -f $lib ] || continue
  echo "# $lib" >> $O/r04_conc_final.txt
  for cfg in "g1msm 128" "g2msm 128" "pairing 8" "pairing 16"; do
    for T in 1 16 64; do timeout -k 10 120 /tmp/conc_bench $lib $cfg $T 60 2>/dev/null >> $O/r04_conc_final.txt; done
  done
done
step degenerate;       timeout -k 10 300 python tools/degenerate_timing.py 2>&1 | grep -v amdgpu.ids > $O/r04_degenerate_inputs.txt
step 4 ranks over gloo; BENCH_DIST_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 4 --steps 5 --warmup 2 > $O/r04_bench_4rank_gloo_rehearsal.json 2> $O/bench_4rank.err || exit 1
step fuzz;             timeout -k 10 400 python tools/fuzz_long.py --seconds 180 --threads 4 > $O/r04_fuzz_long.txt 2>&1 || exit 1
step part 2 done
fi
