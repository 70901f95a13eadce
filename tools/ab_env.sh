#!/bin/bash
# usage: ab_env.sh "<lib> <ENV=val ...>" ...   one bench line per configuration (WL=g1msm by default)
for cfg in "$@"; do
  set -- $cfg; L=$1; shift
  env EIP2537_HIP_LIB=$PWD/variants/$L "$@" timeout -k 10 200 python bench.py --workload ${WL:-g1msm} --log2n ${LOG2N:-20} --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/abe_tmp.json 2>gpurun_out/abe_tmp.err || { echo "$cfg FAILED"; tail -3 gpurun_out/abe_tmp.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/abe_tmp.json')); print('%-50s'%'$cfg', 'ms/step %.3f'%d['ms_per_step'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'dominant %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
done
