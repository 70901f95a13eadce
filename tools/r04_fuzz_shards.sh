export GPU_MAX_HW_QUEUES=16
O=gpurun_out/fz; mkdir -p $O
EIP2537_H2D_STAGES=1,2,3,1 timeout -k 10 400 python tools/fuzz_staged.py --cases 10 --seed 11 2>&1 | grep -v amdgpu.ids | tee $O/staged.txt | tail -3
EIP2537_DEV_STAGES=1,2,3,1 timeout -k 10 400 python tools/fuzz_staged.py --cases 10 --seed 12 --dev 2>&1 | grep -v amdgpu.ids | tee $O/dev.txt | tail -3
EIP2537_DEV_STAGES=5 timeout -k 10 400 python tools/fuzz_staged.py --cases 8 --seed 13 --dev 2>&1 | grep -v amdgpu.ids | tee -a $O/dev.txt | tail -2
