#!/usr/bin/env python3
"""ns/op, gas/op and mgas/s of every precompile through the plain host-buffer C-ABI -- the unit
the reference's Go bench reports (go/blst_eip2537_test.go:112-130: mgas/s = 1000 * gas / ns; the
input is copied per iteration there, here the caller's bytes are staged H2D inside the call).

Sizes follow the reference's Rust bench (rust/benches/eip2537_benches.rs:69-70,134,177: 2..4096
pairs) plus the north-star sizes.  Gas comes from the library's own bls12_*_gas symbols
(src/eip2537.c:1168-1271).  Results are checked for being identical across iterations only; parity
is the test suite's job.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blst_eip2537_amd import Eip2537Executor as X  # noqa: E402


def timed(fn, inp, min_iters=5, budget_s=0.5):
    want = fn(inp)
    fn(inp)
    n, t0 = 0, time.perf_counter()
    while True:
        assert fn(inp) == want
        n += 1
        dt = time.perf_counter() - t0
        if n >= min_iters and dt > budget_s:
            return dt / n * 1e9


def row(name, gas, ns, units=None):
    extra = "" if units is None else "  %10.3g pairs/s" % (units / (ns * 1e-9))
    print("%-26s %12d gas/op %14.0f ns/op %10.2f mgas/s%s" % (name, gas, ns, 1000.0 * gas / ns, extra), flush=True)


def main():
    g1 = X.gen_msm_input("g1", 2, 3, 5, 7)
    g2 = X.gen_msm_input("g2", 2, 3, 5, 7)
    row("g1add", X.gas("g1add"), timed(X.g1_add, g1[:128] + g1[160:288]))
    row("g1mul", X.gas("g1mul"), timed(X.g1_mul, g1[:160]))
    row("g2add", X.gas("g2add"), timed(X.g2_add, g2[:256] + g2[288:544]))
    row("g2mul", X.gas("g2mul"), timed(X.g2_mul, g2[:288]))
    row("map_fp_to_g1", X.gas("map_fp_to_g1"), timed(X.map_fp_to_g1, bytes(16) + bytes(range(1, 49))))
    row("map_fp2_to_g2", X.gas("map_fp2_to_g2"), timed(X.map_fp2_to_g2, (bytes(16) + bytes(range(1, 49))) * 2))
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096, 1 << 16, 1 << 20):
        inp = X.gen_msm_input("g1", n, 3, 5, 0x25370000 + n)
        row("g1multiexp n=%d" % n, X.gas("g1multiexp", len(inp)), timed(X.g1_multiexp, inp), n)
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096, 1 << 16):
        inp = X.gen_msm_input("g2", n, 3, 5, 0x25370000 + n)
        row("g2multiexp n=%d" % n, X.gas("g2multiexp", len(inp)), timed(X.g2_multiexp, inp), n)
    for k in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096):
        inp = X.gen_pairing_input(k, 3, 5, 7, 11)
        row("pairing k=%d" % k, X.gas("pairing", len(inp)), timed(X.pairing, inp), k)


if __name__ == "__main__":
    main()
