#!/bin/bash
# round 4, third GPU call: fused sort-stage kernels / runtime slices / lazy streams: correctness, small-call A/B, stage sweep, size sweep
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04c; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
gcc -O2 tools/conc_bench.c -ldl -lpthread -o /tmp/conc_bench
for lib in variants/libeip2537_hip_r2.so blst_eip2537_amd/libeip2537_hip.so; do
  echo "# $lib" >> $O/conc.txt
  for cfg in "g1msm 128" "g2msm 128" "pairing 8" "pairing 16"; do
    for T in 1 16 64; do
      timeout -k 10 120 /tmp/conc_bench $lib $cfg $T 60 2>/dev/null >> $O/conc.txt
    done
  done
done
cat $O/conc.txt
for st in off default 4 5 "1,3,4,4,4" "1,2,3,3,3,4" "1,3,3,3,3,3" "1,2,2,2,2,2,2,3"; do
  if [ "$st" = off ]; then EIP2537_H2D_PIPELINE=0 timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  elif [ "$st" = default ]; then timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  else EIP2537_H2D_STAGES=$st timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1; fi
done
grep -v amdgpu.ids $O/stages.txt
bash tools/sweep_sizes.sh > $O/sweep.txt 2>&1; cat $O/sweep.txt
timeout -k 10 200 python tools/degenerate_timing.py > $O/degenerate.txt 2>&1; grep -v amdgpu.ids $O/degenerate.txt
