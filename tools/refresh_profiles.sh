#!/bin/bash
# refresh the round's measurement artefacts (run on the GPU box from the repo root)
set -e
R=$PWD
python bench.py > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err
python bench.py --workload g2msm --steps 10 --warmup 2 > gpurun_out/bench_g2.json 2>> gpurun_out/bench_latest.err
python tools/sweep.py > gpurun_out/sweep.txt 2>&1
python tools/gas_bench.py > gpurun_out/mgas.txt 2>&1
python tools/concurrency_timing.py > gpurun_out/conc.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_default -- python3 $R/bench.py > $R/gpurun_out/ks_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_pairing -- python3 $R/bench.py --workload pairing --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ks_pairing.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_g2 -- python3 $R/bench.py --workload g2msm --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ks_g2.log 2>&1
cd $R
ls gpurun_out/ks_default/*/ gpurun_out/ks_pairing/*/ gpurun_out/ks_g2/*/
