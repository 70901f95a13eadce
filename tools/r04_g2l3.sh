export GPU_MAX_HW_QUEUES=16
O=gpurun_out/g2l3; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q failed $O/pytest.log && exit 1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 40 --threads 4 2>&1 | tail -1
bash tools/ab_g2_limb.sh "$@" 
cp gpurun_out/g2l/ab.txt $O/ab.txt
for v in 0 1; do echo "EIP2537_G2_LIMB=$v"; EIP2537_G2_LIMB=$v timeout -k 10 300 python tools/degenerate_timing.py 2>&1 | grep -v amdgpu.ids; done | tee $O/degenerate.txt
