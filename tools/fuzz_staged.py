#!/usr/bin/env python3
"""Differential fuzz of the STAGED host-input path (round 4: record shards over one bucket space, csrc/msm.hip) against the CPU oracle.
Run on the GPU box with a forced cut, e.g.  EIP2537_H2D_STAGES=1,2,3,1 python tools/fuzz_staged.py --cases 20
Inputs: 2^17 + k records (the smallest size whose plan shares buckets), the library's synthetic points with mutations that cross shard
boundaries -- the same point with the same scalar in several shards (doubling inside a bucket accumulator), a point and its negative in
different shards (an accumulator falls back to infinity and is picked up again), runs of infinities, tiny and huge scalars, a bad record
at a random position (error order across shards)."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402,F401
from oracle import clib  # noqa: E402
import bls12_381 as m  # noqa: E402
from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=12)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--dev", action="store_true", help="device-resident input (eip2537_hip_g1multiexp_dev): the record shards of EIP2537_DEV_STAGES, "
                "sort stage of shard s + 1 beside the accumulate of shard s")
args = ap.parse_args()
rng = random.Random(args.seed)
A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321


def call(inp):
    try:
        if args.dev:
            import torch
            d = torch.frombuffer(bytearray(inp), dtype=torch.uint8).cuda()
            return 0, X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), len(inp) // 160)
        return 0, X.g1_multiexp(inp)
    except Eip2537Error as e:
        return e.code, None


bad = 0
t0 = time.time()
for case in range(args.cases):
    n = (1 << 17) + rng.randrange(1, 6000)
    buf = bytearray(X.gen_msm_input("g1", n, A, B, rng.randrange(1 << 30)))
    rec = lambda i: buf[i * 160:(i + 1) * 160]
    ks = [0, 1, 2, 0xffff, 0x10000, m.R - 1, m.R, 2 ** 255, 2 ** 256 - 1, rng.randrange(1 << 256)]
    for _ in range(rng.randrange(0, 40)):                       # the same (point, scalar) at random places all over the input
        src = rng.randrange(n)
        r = bytes(rec(src)[:128]) + m.encode_scalar(rng.choice(ks))
        for _ in range(rng.randrange(1, 6)):
            d = rng.randrange(n)
            buf[d * 160:(d + 1) * 160] = r
    for _ in range(rng.randrange(0, 40)):                       # a point here, its negative with the same scalar somewhere else
        src, d = rng.randrange(n), rng.randrange(n)
        x, y = buf[src * 160:src * 160 + 64], int.from_bytes(buf[src * 160 + 80:src * 160 + 128], "big")
        buf[d * 160:(d + 1) * 160] = bytes(x) + bytes(16) + ((m.P - y) % m.P).to_bytes(48, "big") + bytes(buf[src * 160 + 128:src * 160 + 160])
    for _ in range(rng.randrange(0, 4)):                        # runs of infinities
        a = rng.randrange(n)
        ln = min(n - a, rng.randrange(1, 3000))
        for i in range(a, a + ln):
            buf[i * 160:i * 160 + 128] = bytes(128)
    if rng.random() < 0.4:                                       # one or two bad records: the lowest index wins
        for _ in range(rng.randrange(1, 3)):
            i = rng.randrange(n)
            if rng.random() < 0.5:
                buf[i * 160] = 1                                 # pad byte
            else:
                buf[i * 160 + 16:i * 160 + 128] = m.encode_g1((1, 1))[16:]       # off the curve
    inp = bytes(buf)
    got, want = call(inp), clib.call("bls12_g1multiexp", inp)
    ok = got == want
    bad += 0 if ok else 1
    print("case %2d n=%d shards=%s rc=%d %s (%.0f s)" % (case, n, (X.last_plan() or {}).get("shards"), want[0], "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
print("fuzz_staged: %d cases, %s, MISMATCHES: %d" % (args.cases, ("EIP2537_DEV_STAGES=%s (device-resident)" % os.environ.get("EIP2537_DEV_STAGES")) if args.dev
      else "EIP2537_H2D_STAGES=%s" % os.environ.get("EIP2537_H2D_STAGES"), bad))
sys.exit(1 if bad else 0)
