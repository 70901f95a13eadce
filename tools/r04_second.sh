#!/bin/bash
# round 4, second GPU call: full GPU suite, timeline of the staged host call, bisect of the small-call regression, bench line, traces
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04b; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_host -- python3 $R/tools/dbg_host_abi.py 20 g1 8 > $O/trace_host.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_g1_16 -- python3 $R/bench.py --workload g1msm --log2n 16 --steps 4 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0 > $O/trace_g1_16.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_pair -- python3 $R/bench.py --workload pairing --steps 4 --warmup 2 --no-cpu-baseline --no-host-abi --sustained 0 > $O/trace_pair.log 2>&1
cd $R
python3 tools/trace_call.py $O/trace_host k_msm_reduce_rc 7 > $O/timeline_host.txt 2>&1
python3 tools/trace_call.py $O/trace_g1_16 k_msm_reduce4 1.2 > $O/timeline_g1_16.txt 2>&1
python3 tools/trace_call.py $O/trace_pair k_pair_tree2 1.5 > $O/timeline_pair.txt 2>&1
rm -rf $O/trace_host $O/trace_g1_16 $O/trace_pair
gcc -O2 tools/conc_bench.c -ldl -lpthread -o /tmp/conc_bench
for lib in variants/libeip2537_hip_r2.so variants/libeip2537_hip_a.so variants/libeip2537_hip_b.so variants/libeip2537_hip_d.so variants/libeip2537_hip_e.so blst_eip2537_amd/libeip2537_hip.so; do
  [ -f $lib ] || continue
  echo "# $lib" >> $O/bisect.txt
  for cfg in "g1msm 128" "g2msm 128"; do
    for T in 16 64; do
      timeout -k 10 120 /tmp/conc_bench $lib $cfg $T 60 2>/dev/null >> $O/bisect.txt
    done
  done
done
cat $O/bisect.txt
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','bit_exact_vs_golden')}, d['step_ms'])
print('host_abi', d['host_abi']['ms_per_call'], d['host_abi']['matches_device_resident_result'], d['host_abi']['plan'])
print('sustained', d['sustained'])
print('roofline_valu', d['roofline_valu'])
s=d['secondary']; print('pairing', s['ms_per_check'], s.get('sustained'), s['host_abi']['ms_per_call'])
"
/opt/rocm/bin/hipcc -O3 -frounding-math --offload-arch=gfx950 -I blst_eip2537_amd/csrc tools/fpmul_bench.hip -o /tmp/fpmul_bench 2>/dev/null && /tmp/fpmul_bench > $O/fpmul_bench.txt; cat $O/fpmul_bench.txt
