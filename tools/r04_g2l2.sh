export GPU_MAX_HW_QUEUES=16
for lib in variants/libeip2537_hip_g2l_w2p0.so variants/libeip2537_hip_g2l_w1p1.so; do
  EIP2537_HIP_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "g2 or G2 or config3" 2>&1 | tail -1
done
bash tools/ab_g2_limb.sh variants/libeip2537_hip_g2l_w2p0.so variants/libeip2537_hip_g2l_w1p1.so
