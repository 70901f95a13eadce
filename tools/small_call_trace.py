#!/usr/bin/env python3
"""Issue a few small precompile calls (run under rocprofv3 --kernel-trace to see their kernel timelines)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blst_eip2537_amd import Eip2537Executor as X  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "pairing"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if what == "pairing":
    inp, fn = X.gen_pairing_input(n, 3, 5, 7, 11), X.pairing
elif what == "g1msm":
    inp, fn = X.gen_msm_input("g1", n, 3, 5, 7), X.g1_multiexp
else:
    inp, fn = X.gen_msm_input("g2", n, 3, 5, 7), X.g2_multiexp
for _ in range(3):
    fn(inp)
t0 = time.perf_counter()
for _ in range(5):
    fn(inp)
print("%s n=%d: %.3f ms per call, device pipeline %.3f ms" % (what, n, (time.perf_counter() - t0) / 5 * 1e3, X.last_timing()[0]))
