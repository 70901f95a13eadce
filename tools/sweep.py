#!/usr/bin/env python3
"""Size sweep of the three hot paths (SURVEY.md 8d "scale sweep"): runs bench.py per size and prints one line each."""
import json
import subprocess
import sys

CASES = [("g1msm", l) for l in (8, 10, 12, 14, 16, 18, 20)] + [("g2msm", l) for l in (8, 10, 12, 14, 16)] + \
        [("pairing", l) for l in (4, 6, 8, 10, 12)]
for wl, l in CASES:
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--workload", wl, "--log2n", str(l),
                          "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True).stdout
    line = [x for x in out.splitlines() if x.startswith("{")]
    if not line:
        print(wl, l, "FAILED")
        continue
    d = json.loads(line[0])
    print("%-8s 2^%-2d  %8.3f ms/call  %10.3g pairs/s   dominant kernel %7.3f ms  device pipeline %7.3f ms  ok=%s"
          % (wl, l, d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"], d["roofline"]["device_pipeline_ms"], d["bit_exact_vs_golden"]), flush=True)
