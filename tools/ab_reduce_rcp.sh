# A/B of the two-level bucket reduce of the c = 13 plans (k_msm_rowcol_p + k_msm_reduce_rc_p) against the chain reduce (EIP2537_REDUCE_RCP=0).
export GPU_MAX_HW_QUEUES=16
O=gpurun_out/rcp; mkdir -p $O
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q passed $O/pytest.log || exit 1
grep -q failed $O/pytest.log && exit 1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 60 --threads 4 2>&1 | tail -3
timeout -k 10 200 python tools/fuzz_long.py --window 13 --seconds 40 --threads 4 2>&1 | tail -3
one() { python bench.py --workload g1msm --log2n $1 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi --sustained 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('2^$1 RCP=${EIP2537_REDUCE_RCP:-1}', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%r['device_pipeline_ms'], 'reduce %.3f'%r.get('fold_reduce_ms',0), 'exact', d['bit_exact_vs_golden'])"; }
for rep in 1 2; do for l in 14 16 17; do
  EIP2537_REDUCE_RCP=0 one $l; one $l
done; done | tee $O/ab.txt
for v in 0 1; do EIP2537_REDUCE_RCP=$v timeout -k 10 120 python tools/dbg_host_abi.py 16 g1 2>&1 | grep -v amdgpu.ids; done | tee -a $O/ab.txt
