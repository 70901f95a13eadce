#!/usr/bin/env python3
"""Aggregate call rate of the reference ABI under concurrent callers (SURVEY.md 8b "Threading").

T host threads each issue M calls of a small precompile input through the plain C-ABI
(ctypes releases the GIL for the duration of the call).  Run once per EIP2537_HIP_SLOTS value
(the pool size is read when the library first selects its device):

    EIP2537_HIP_SLOTS=1 python tools/concurrency_timing.py
    EIP2537_HIP_SLOTS=4 python tools/concurrency_timing.py
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blst_eip2537_amd import Eip2537Executor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32, 64])
    ap.add_argument("--calls", type=int, default=40)
    args = ap.parse_args()
    ex = Eip2537Executor()
    work = {
        "g1msm_128": ("g1_multiexp", ex.gen_msm_input("g1", 128, 3, 5, 7)),
        "g1msm_4096": ("g1_multiexp", ex.gen_msm_input("g1", 4096, 3, 5, 7)),
        "g2msm_128": ("g2_multiexp", ex.gen_msm_input("g2", 128, 3, 5, 7)),
        "pairing_2": ("pairing", ex.gen_pairing_input(2, 3, 5, 7, 11)),
        "pairing_8": ("pairing", ex.gen_pairing_input(8, 3, 5, 7, 11)),
        "pairing_16": ("pairing", ex.gen_pairing_input(16, 3, 5, 7, 11)),
    }
    slots = os.environ.get("EIP2537_HIP_SLOTS", "8 (default)")
    print("slots=%s coalesce=%s" % (slots, os.environ.get("EIP2537_HIP_COALESCE", "1 (default)")))
    for name, (fn, inp) in work.items():
        want = getattr(ex, fn)(inp)           # warm-up, and the value every thread must reproduce
        for _ in range(8):
            getattr(ex, fn)(inp)
        row = []
        for t in args.threads:
            bad = []

            def run():
                for _ in range(args.calls):
                    if getattr(ex, fn)(inp) != want:
                        bad.append(1)

            # untimed round first: slots touched for the first time create their streams and
            # grow their workspaces (hipMalloc synchronises the device)
            calls, args.calls = args.calls, 4
            th = [threading.Thread(target=run) for _ in range(t)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            args.calls = calls
            th = [threading.Thread(target=run) for _ in range(t)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - t0
            assert not bad, "result mismatch under concurrency"
            row.append("T=%d %7.0f calls/s" % (t, t * args.calls / dt))
        print("%-12s %s" % (name, "  ".join(row)))
    print("coalesce stats (device pipelines, calls, largest batch): %s" % (ex.coalesce_stats(),))


if __name__ == "__main__":
    main()
