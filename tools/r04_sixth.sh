#!/bin/bash
# round 4, sixth GPU call: sort stage of shard s+1 beside the accumulate of shard s: correctness, schedules
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04f; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
timeout -k 10 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
run() { timeout -k 10 120 python tools/dbg_host_abi.py ${2:-20} g1 >> $O/stages.txt 2>&1; }
EIP2537_H2D_OVERLAP=0 run
run
for st in "1,4,4,4,3" "2,6,5,4,3,2" "1,5,4,3,2,1" "2,8,6,4,2,1" "3,12,10,8,6,5,4,3" "1,4,4,3,2,1,1" "2,6,6,5,3,1" "1,6,5,3,1"; do
  EIP2537_H2D_STAGES=$st run
done
for l in 19 21 22; do EIP2537_H2D_OVERLAP=0 run x $l; run x $l; done
grep -v amdgpu.ids $O/stages.txt
cd /tmp && export TMPDIR=/tmp
EIP2537_H2D_STAGES=2,6,5,4,3,2 timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_host -- python3 $R/tools/dbg_host_abi.py 20 g1 8 > $O/trace_host.log 2>&1
cd $R
python3 tools/trace_call.py $O/trace_host k_msm_reduce_rc 6 > $O/timeline_host.txt 2>&1
rm -rf $O/trace_host
tail -100 $O/timeline_host.txt
