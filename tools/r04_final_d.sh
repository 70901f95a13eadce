export GPU_MAX_HW_QUEUES=16
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_long.py --seconds 120 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --mid --seconds 45 --threads 4 2>&1 | tail -1
timeout -k 10 200 python tools/fuzz_long.py --window 16 --seconds 30 --threads 4 2>&1 | tail -1
python tools/time_host_ops.py; EIP2537_HOST_IFMA=0 python tools/time_host_ops.py
timeout -k 10 300 python bench.py --steps 10 --warmup 3 | cut -c1-400
