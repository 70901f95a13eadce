export GPU_MAX_HW_QUEUES=16
bash tools/refresh_r04.sh || exit 1
rm -f gpurun_out/final4/r04_host_abi_sizes.txt
bash tools/refresh_r04.sh part3 || exit 1
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
