#!/usr/bin/env python3
"""Small-call crossover table: time of one reference-ABI call through the product (default
route: the library's host code below the crossover, the GPU above it; also with the route pinned) against
the 1-core CPU port of the reference path (oracle/), for the sizes the EVM actually sends.  Run on the GPU
box; output goes to profiles/rNN_small_calls.txt."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402,F401
from oracle import clib  # noqa: E402
from blst_eip2537_amd import Eip2537Executor as X  # noqa: E402

A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321


def med_us(fn, arg, reps):
    fn(arg)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn(arg)
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts)


def row(name, n, fn, oname, inp):
    reps = 15 if n <= 64 else 7
    X.set_route(-1)
    d = med_us(fn, inp, reps)
    X.set_route(0)
    g = med_us(fn, inp, reps)
    X.set_route(1)
    h = med_us(fn, inp, reps) if n <= 64 else float("nan")
    X.set_route(-1)
    p = med_us(lambda b: clib.call(oname, b), inp, 3 if n > 16 else 7)
    same = fn(inp) == clib.call(oname, inp)[1]
    print("%-8s %5d  product %9.0f us (%7.0f ns/unit)   gpu-route %9.0f   host-route %9.0f   port(1 core) %10.0f   port/product %5.1fx  %s"
          % (name, n, d, d * 1e3 / n, g, h, p, p / d, "bit-exact" if same else "MISMATCH"))
    return p / d


def main():
    print("# one call at a time, median of repeated calls, microseconds; host: %d cores visible" % len(os.sched_getaffinity(0)))
    worst = 1e9
    for n in (1, 2, 3, 4, 5, 8, 16, 32, 64, 128, 256):
        worst = min(worst, row("g1msm", n, X.g1_multiexp, "bls12_g1multiexp", clib.gen_msm_input("g1", n, A, B, n)))
    for n in (1, 2, 3, 4, 5, 8, 16, 32, 64, 128):
        worst = min(worst, row("g2msm", n, X.g2_multiexp, "bls12_g2multiexp", clib.gen_msm_input("g2", n, A, B, n)))
    for k in (1, 2, 3, 4, 8, 16, 32, 64):
        worst = min(worst, row("pairing", k, X.pairing, "bls12_pairing", clib.gen_pairing_input(k, 5, 7, 11, 13)))
    print("# smallest port/product ratio over the table: %.2fx (>= 1 means the product is never slower than the port)" % worst)


if __name__ == "__main__":
    main()
