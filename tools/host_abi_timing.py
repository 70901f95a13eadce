#!/usr/bin/env python3
"""Latency of the reference ABI calls with a HOST buffer (what the Rust/Go bindings pass), i.e.
including the host-to-device copy, against the device-resident entry point bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blst_eip2537_amd import Eip2537Executor as X
A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321
for log2n in (12, 16, 20):
    n = 1 << log2n
    host = X.gen_msm_input("g1", n, A, B, 0x25370000 + log2n)
    d = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    out_d = X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    assert X.g1_multiexp(host) == out_d
    t = time.perf_counter()
    for _ in range(5): X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    td = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    for _ in range(5): X.g1_multiexp(host)
    th = (time.perf_counter() - t) / 5
    print("g1 msm 2^%d: device-resident %.2f ms, host buffer (bls12_g1multiexp) %.2f ms  (+%.2f ms for %.1f MB = %.1f GB/s)"
          % (log2n, td * 1e3, th * 1e3, (th - td) * 1e3, len(host) / 1e6, len(host) / 1e9 / max(th - td, 1e-9)), flush=True)
