#!/bin/bash
# round 4, first GPU call: correctness of the staged pipeline, stage sweep, native concurrency A/B, sustained VALU probe
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04a; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_split.py tests/test_gpu_routes.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for st in off 2 3 4 5 6 "1,4,4,4,4" "1,3,4,4,4" "1,2,3,3,3,4" "2,4,4,4,3" "1,4,4,4,4,2" default; do
  if [ "$st" = off ]; then EIP2537_H2D_PIPELINE=0 timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  elif [ "$st" = default ]; then timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  else EIP2537_H2D_STAGES=$st timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1; fi
done
for l in 18 19 21 22; do
  EIP2537_H2D_PIPELINE=0 timeout -k 10 120 python tools/dbg_host_abi.py $l g1 13 >> $O/stages.txt 2>&1
  timeout -k 10 120 python tools/dbg_host_abi.py $l g1 13 >> $O/stages.txt 2>&1
done
EIP2537_H2D_STAGES=1,3 timeout -k 10 120 python tools/dbg_host_abi.py 18 g1 13 >> $O/stages.txt 2>&1
EIP2537_H2D_STAGES=1,2,2 timeout -k 10 120 python tools/dbg_host_abi.py 18 g1 13 >> $O/stages.txt 2>&1
grep -v amdgpu.ids $O/stages.txt
# native concurrency: final binary against the round-2 library, same box
gcc -O2 tools/conc_bench.c -ldl -lpthread -o /tmp/conc_bench
for lib in blst_eip2537_amd/libeip2537_hip.so variants/libeip2537_hip_r2.so; do
  for co in 1 0; do
    echo "# $lib EIP2537_HIP_COALESCE=$co" >> $O/conc.txt
    for cfg in "g1msm 128" "g2msm 128" "pairing 8" "pairing 16"; do
      for T in 1 16 64; do
        EIP2537_HIP_COALESCE=$co timeout -k 10 120 /tmp/conc_bench $lib $cfg $T 60 2>/dev/null >> $O/conc.txt
      done
    done
  done
done
cat $O/conc.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe > $O/valu_probe.txt && /tmp/valu_probe 2.0 >> $O/valu_probe.txt
tail -12 $O/valu_probe.txt
nproc; lscpu | grep "Model name"
