# two-level bucket reduce for the c = 8 plans (up to 2 048 records), G1 and G2, against the running-sum chains (EIP2537_REDUCE_RCP=0 / EIP2537_REDUCE_RCP8=0)
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/rcps; mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q failed $O/pytest.log && exit 1
timeout -k 10 200 python tools/fuzz_long.py --seconds 90 --threads 4 2>&1 | tail -1
EIP2537_HIP_COALESCE=0 timeout -k 10 200 python tools/fuzz_long.py --seconds 60 --threads 4 --seed 7 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --window 8 --seconds 45 --threads 4 2>&1 | tail -1
one() { python bench.py --workload $1 --log2n $2 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$2 RCP=${EIP2537_REDUCE_RCP:-1} RCP8=${EIP2537_REDUCE_RCP8:-1}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0), 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 5 7 10 11; do
  EIP2537_REDUCE_RCP=0 one g1msm $l; one g1msm $l; EIP2537_REDUCE_RCP8=0 one g2msm $l; one g2msm $l
done; done | tee $O/ab.txt
