#!/bin/bash
# round 4, twelfth GPU call: timeline of the final binary's staged host call; a last look at the cut
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04l; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_host -- python3 $R/tools/dbg_host_abi.py 20 g1 8 > $O/trace_host.log 2>&1
cd $R
python3 tools/trace_call.py $O/trace_host k_msm_reduce_rc 5.2 > $O/timeline_host.txt 2>&1
rm -rf $O/trace_host
grep -c . $O/timeline_host.txt
for st in default "1,3,3,3,3,3" "1,5,5,5" "1,4,4,4,3" "2,7,7,7,7"; do
  if [ "$st" = default ]; then timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1
  else EIP2537_H2D_STAGES=$st timeout -k 10 120 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1; fi
done
grep -v amdgpu.ids $O/stages.txt
