export GPU_MAX_HW_QUEUES=16
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_long.py --seconds 90 --threads 4 2>&1 | tail -1
timeout -k 10 300 python tools/small_calls.py 2>&1 | grep "pairing  *[1-4] \|g2msm  *[1-8] "
