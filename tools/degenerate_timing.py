#!/usr/bin/env python3
"""Timing of degenerate MSM inputs (all scalars equal; all points equal) on the GPU: these put
every record of a window into one bucket.  Correctness is covered by tests/test_gpu_fullsize.py."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blst_eip2537_amd import Eip2537Executor as X
A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321
n = 1 << 18
base = X.gen_msm_input("g1", n, A, B, 5)
k = (2 ** 256 - 1).to_bytes(32, "big")
cases = {"random (reference)": base,
         "all scalars equal": b"".join(base[i * 160:i * 160 + 128] + k for i in range(n)),
         "all records equal": (base[:128] + k) * n}
for name, inp in cases.items():
    d = torch.frombuffer(bytearray(inp), dtype=torch.uint8).cuda()
    X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    t = time.perf_counter()
    for _ in range(3):
        X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    print("%-22s n=2^18: %.2f ms per call" % (name, (time.perf_counter() - t) / 3 * 1e3), flush=True)

# G2 (round 4): all scalars equal sends a whole window to ONE bucket, i.e. through k_msm_fold_big<Fp2, Fp2> -- the one kernel with a
# large scratch frame (944 B per lane) that an adversarial input can reach
n2 = 1 << 16
base2 = X.gen_msm_input("g2", n2, A, B, 5)
cases2 = {"G2 random (reference)": base2,
          "G2 all scalars equal": b"".join(base2[i * 288:i * 288 + 256] + k for i in range(n2)),
          "G2 all records equal": (base2[:256] + k) * n2}
for name, inp in cases2.items():
    d = torch.frombuffer(bytearray(inp), dtype=torch.uint8).cuda()
    X.dev_call("eip2537_hip_g2multiexp_dev", d.data_ptr(), n2)
    t = time.perf_counter()
    for _ in range(3):
        X.dev_call("eip2537_hip_g2multiexp_dev", d.data_ptr(), n2)
    print("%-22s n=2^16: %.2f ms per call" % (name, (time.perf_counter() - t) / 3 * 1e3), flush=True)
