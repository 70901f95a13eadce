#!/usr/bin/env python3
"""Long differential fuzz of the GPU paths against the CPU oracle (run on the GPU box; too long for pytest).

Unlike tests/test_gpu_fuzz.py this generator draws points and scalars from SMALL pools, so that coincidences
are the rule: the same point many times in one bucket, a point and its negative, bucket sums that are equal,
opposite or infinity when the reduce adds them, windows whose digits collide -- the complete-addition branches
(P = Q, P = -Q, infinity) of every lane-group point operation (madd2 / madd2c, add4, add8c, the folds).  Calls are
issued from several threads at once, so they also ride through the coalescing queue.
    python tools/fuzz_long.py [--seconds 240] [--threads 4] [--seed 1]
"""
import argparse
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402,F401
from oracle import clib  # noqa: E402
import bls12_381 as m  # noqa: E402
from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error  # noqa: E402


def call(fn, inp):
    try:
        return 0, fn(inp)
    except Eip2537Error as e:
        return e.code, None


def pools(rng):
    prng = m.SplitMix64(rng.randrange(1 << 32))
    g1 = [m.g1_mul(m.G1, rng.randrange(1, 50)) for _ in range(3)]
    g1 += [m.ec_neg(m.FP, p) for p in g1] + [None, (0, 2), m.random_g1(prng, False)]
    g2 = [m.g2_mul(m.G2, rng.randrange(1, 50)) for _ in range(3)]
    g2 += [m.ec_neg(m.FP2, p) for p in g2] + [None, m.random_g2(prng, False)]
    ks = [0, 1, 2, 3, 255, 256, 257, 0x8000, 0x8001, 0xffff, 0x10000, 0x10001, m.R - 1, m.R, m.R + 1, 2 ** 255, 2 ** 256 - 1,
          (1 << 128) + 1, rng.randrange(1 << 256), rng.randrange(1 << 256)]
    if MID:                                           # enough distinct scalars that most buckets of a window are hit
        ks += [rng.randrange(1 << 256) for _ in range(40)]
    return [m.encode_g1(p) for p in g1], [m.encode_g2(p) for p in g2], [m.encode_scalar(k) for k in ks]


MID = False


def gen_case(rng, e1, e2, ks):
    kind = rng.random()
    if MID:                                           # the c = 11 / c = 13 plans: split buckets, folds, 4- and 8-lane reduces
        if kind < 0.6:
            n = rng.choice([2049, 2500, 4096, 7000, 8192, 8193, 12000, 20000])
            return "bls12_g1multiexp", X.g1_multiexp, b"".join(rng.choice(e1) + rng.choice(ks) for _ in range(n))
        n = rng.choice([2049, 2500, 4096, 6000, 8193])
        return "bls12_g2multiexp", X.g2_multiexp, b"".join(rng.choice(e2) + rng.choice(ks) for _ in range(n))
    if kind < 0.45:
        n = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 31, 64, 65, 129, 300, 513, 600])
        return "bls12_g1multiexp", X.g1_multiexp, b"".join(rng.choice(e1) + rng.choice(ks) for _ in range(n))
    if kind < 0.8:
        n = rng.choice([1, 2, 3, 4, 5, 7, 8, 33, 64, 65, 130, 300])
        return "bls12_g2multiexp", X.g2_multiexp, b"".join(rng.choice(e2) + rng.choice(ks) for _ in range(n))
    k = rng.choice([1, 2, 3, 4, 5, 6, 8, 16, 33])
    in_g1 = [p for p in e1[:7]]                       # subgroup points and infinity: a passing decode most of the time
    in_g2 = [q for q in e2[:7]]
    inp = b"".join((rng.choice(e1) if rng.random() < 0.05 else rng.choice(in_g1)) +
                   (rng.choice(e2) if rng.random() < 0.05 else rng.choice(in_g2)) for _ in range(k))
    return "bls12_pairing", X.pairing, inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mid", action="store_true", help="mid-size MSMs (2 049 .. 20 000 records) instead of small calls")
    ap.add_argument("--window", type=int, default=0, help="force the Pippenger window width of the GPU plans (16: the limb-form "
                    "kernels of the large sizes on these small inputs; the width applies to G1 and G2)")
    args = ap.parse_args()
    global MID
    MID = args.mid
    if args.window:
        X.set_window(args.window)
    stop = time.time() + args.seconds
    lock = threading.Lock()
    stats = {"cases": 0, "bad": [], "by_op": {}, "errors_seen": {}}

    def work(tid):
        rng = random.Random(args.seed * 1000 + tid)
        e1, e2, ks = pools(rng)
        i = 0
        while time.time() < stop:
            if i % 50 == 49:
                e1, e2, ks = pools(rng)
            i += 1
            name, fn, inp = gen_case(rng, e1, e2, ks)
            if rng.random() < 0.1 and len(inp) > 64:                      # a pad byte somewhere: INVALID_ELEMENT of the lowest such record
                b = bytearray(inp)
                b[rng.randrange(len(inp) // 64) * 64 + rng.randrange(16)] ^= rng.randrange(1, 256)
                inp = bytes(b)
            want = clib.call(name, inp)
            got = call(fn, inp)
            with lock:
                stats["cases"] += 1
                stats["by_op"][name] = stats["by_op"].get(name, 0) + 1
                stats["errors_seen"][want[0]] = stats["errors_seen"].get(want[0], 0) + 1
                if got != want:
                    stats["bad"].append((name, len(inp), got[0], want[0], inp.hex()[:200]))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(args.threads)]
    t0 = time.time()
    for t in ths:
        t.start()
    while any(t.is_alive() for t in ths):
        time.sleep(30)
        with lock:
            print("... %d cases, %d mismatches after %.0f s" % (stats["cases"], len(stats["bad"]), time.time() - t0), flush=True)
    for t in ths:
        t.join()
    print("fuzz_long: %d cases in %.0f s on %d threads (seed %d): %s; expected codes seen %s; coalescing %s; MISMATCHES: %d"
          % (stats["cases"], time.time() - t0, args.threads, args.seed, stats["by_op"], dict(sorted(stats["errors_seen"].items())),
             X.coalesce_stats(), len(stats["bad"])))
    for b in stats["bad"][:10]:
        print("  BAD", b)
    sys.exit(1 if stats["bad"] else 0)


if __name__ == "__main__":
    main()
