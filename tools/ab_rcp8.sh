# G2 c = 13: two-level bucket reduce (k_msm_rowcol8_p + k_msm_reduce_rc8_p) against the running-sum chain (EIP2537_REDUCE_RCP8=0)
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/rcp8; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q failed $O/pytest.log && exit 1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 45 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --window 13 --seconds 45 --threads 4 2>&1 | tail -1
one() { python bench.py --workload g2msm --log2n $1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('g2 2^$1 RCP8=${EIP2537_REDUCE_RCP8:-1}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0), 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 14 16 18; do EIP2537_REDUCE_RCP8=0 one $l; one $l; done; done | tee $O/ab.txt
for v in 0 1; do echo "EIP2537_REDUCE_RCP8=$v"; EIP2537_REDUCE_RCP8=$v timeout -k 10 300 python tools/degenerate_timing.py 2>&1 | grep "G2"; done | tee -a $O/ab.txt
