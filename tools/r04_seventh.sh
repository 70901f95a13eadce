#!/bin/bash
# round 4, seventh GPU call: H2D probe (one thread or several?), 4-rank gloo rehearsal on the final binary, small-call table
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04g; mkdir -p $O
export GPU_MAX_HW_QUEUES=16
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/h2d_probe.hip -lpthread -o /tmp/h2d_probe 2>/dev/null && /tmp/h2d_probe > $O/h2d_probe.txt 2>&1; cat $O/h2d_probe.txt
timeout -k 10 200 python tools/dbg_host_abi.py 20 g1 >> $O/stages.txt 2>&1; grep -v amdgpu.ids $O/stages.txt
BENCH_DIST_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 4 --steps 6 --warmup 2 > $O/bench_4rank_gloo.json 2> $O/bench_4rank_gloo.err; echo "4-rank rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench_4rank_gloo.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus','bit_exact_vs_golden')}); print(d.get('strong')); print(d.get('in_library_split'))"
timeout -k 10 500 python tools/small_calls.py > $O/small_calls.txt 2>&1; grep -v amdgpu.ids $O/small_calls.txt | tail -40
