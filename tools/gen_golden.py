#!/usr/bin/env python3
"""Generate tests/golden/* from the Python big-integer model (oracle/pymodel) ONLY.

The reference cannot be run here (its arithmetic is blst, fetched from the network by its
build.sh) and ships no vectors in-tree, so these are goldens of an independent implementation,
not of the reference: parity stays "unpinned" (DESIGN.md).  What they pin is that the C oracle
and the HIP engine agree with a third, structurally different implementation.

  kat.json              known-answer vectors: {op, input, code, output}
  g1msm_2p16.hex ...    analytic goldens of the SURVEY.md 8d workloads: the inputs are
                        P_i = [A + i*B]G with SplitMix64 scalars, so the expected MSM output is
                        the single multiplication [(sum k_i (A + i*B)) mod r]G.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pymodel"))
import bls12_381 as m  # noqa: E402
import h2c  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321


def seed_for(workload, log2n):
    return 0x25370000 + {"g1msm": 0, "g2msm": 0x100}[workload] + log2n


def analytic(workload, log2n):
    rng = m.SplitMix64(seed_for(workload, log2n))
    s = 0
    for i in range(1 << log2n):
        s += rng.scalar256() * ((A + i * B) % m.R)
    s %= m.R
    if workload == "g1msm":
        return m.encode_g1(m.g1_mul(m.G1, s))
    return m.encode_g2(m.g2_mul(m.G2, s))


def kat():
    v = []
    rng = m.SplitMix64(0x2537)
    G1, G2 = m.G1, m.G2
    P, Q = m.g1_mul(G1, 0xabcdef0123), m.g1_mul(G1, 0x9876543210)
    nP = m.ec_neg(m.FP, P)
    e1, e2, es = m.encode_g1, m.encode_g2, m.encode_scalar

    def add(op, fn, inp):
        code, out = m.call(fn, inp)
        v.append({"op": op, "input": inp.hex(), "code": code, "output": out.hex() if out is not None else None})

    # BASELINE config 1: single-pair g1add / g1mul (+ failures)
    for inp in [e1(G1) + e1(G1), e1(G1) + bytes(128), bytes(256), e1(P) + e1(nP), e1(P) + e1(Q),
                e1((0, 2)) + e1((0, 2)), e1((0, 2)) + e1(P),
                bytes(255), b"", e1(P)[:3] + b"\x01" + e1(P)[4:] + e1(Q),
                m.encode_fp(0)[:16] + m.P.to_bytes(48, "big") + m.encode_fp(1) + e1(Q),
                m.encode_fp(1) + m.encode_fp(1) + e1(Q), e1(P) + m.encode_fp(0) + m.encode_fp(2)[:63] + b"\x03"]:
        add("g1add", m.bls12_g1add, inp)
    for k in [0, 1, 2, m.R - 1, m.R, m.R + 1, 2 ** 255, 2 ** 256 - 1, rng.scalar256()]:
        add("g1mul", m.bls12_g1mul, e1(G1) + es(k))
        add("g1mul", m.bls12_g1mul, e1((0, 2)) + es(k))
    add("g1mul", m.bls12_g1mul, bytes(161))
    add("g1mul", m.bls12_g1mul, m.encode_fp(1) + m.encode_fp(1) + es(5))
    S, T = m.g2_mul(G2, 0x13579bdf), m.g2_mul(G2, 0x2468ace0)
    for inp in [e2(G2) + e2(G2), e2(S) + e2(T), e2(S) + e2(m.ec_neg(m.FP2, S)), e2(S) + bytes(256), bytes(511),
                m.encode_fp(1) * 4 + e2(T)]:
        add("g2add", m.bls12_g2add, inp)
    for k in [0, 1, m.R, m.R + 1, 2 ** 256 - 1, rng.scalar256()]:
        add("g2mul", m.bls12_g2mul, e2(G2) + es(k))
    # small MSMs with non-subgroup points and the dispatcher thresholds 1 / <=4 / else
    for n in [1, 2, 4, 5, 9]:
        recs = b"".join(e1(m.random_g1(rng, in_subgroup=(i % 3 != 0))) + es(rng.scalar256()) for i in range(n))
        add("g1multiexp", m.bls12_g1multiexp, recs)
    add("g1multiexp", m.bls12_g1multiexp, e1(P) + es(5) + e1(nP) + es(5) + bytes(160) + e1(Q) + es(0) + e1(P) + es(m.R))
    bad = bytearray(b"".join(e1(m.g1_mul(G1, i + 2)) + es(i + 1) for i in range(6)))
    bad[4 * 160:4 * 160 + 128] = m.encode_fp(1) + m.encode_fp(1)
    add("g1multiexp", m.bls12_g1multiexp, bytes(bad))
    bad[2 * 160 + 5] = 7
    add("g1multiexp", m.bls12_g1multiexp, bytes(bad))
    add("g1multiexp", m.bls12_g1multiexp, bytes(159))
    for n in [1, 3, 5]:
        recs = b"".join(e2(m.random_g2(rng, in_subgroup=(i % 2 == 0))) + es(rng.scalar256()) for i in range(n))
        add("g2multiexp", m.bls12_g2multiexp, recs)
    # pairing checks
    pr = lambda ps: b"".join(e1(p) + e2(q) for p, q in ps)
    ns1, ns2 = m.random_g1(rng, False), m.random_g2(rng, False)
    for inp in [pr([(m.g1_mul(G1, 3), G2), (m.ec_neg(m.FP, G1), m.g2_mul(G2, 3))]),
                pr([(m.g1_mul(G1, 3), G2), (m.ec_neg(m.FP, G1), m.g2_mul(G2, 4))]),
                pr([(G1, G2)]), pr([(None, G2), (G1, None)]), pr([(None, None)]),
                pr([(m.g1_mul(G1, 6), m.g2_mul(G2, 35)), (m.ec_neg(m.FP, m.g1_mul(G1, 10)), m.g2_mul(G2, 21)), (None, G2)]),
                pr([(ns1, G2)]), pr([(G1, ns2)]), pr([(G1, G2), ((1, 1), ns2)]), pr([(ns1, ns2)]),
                bytes(383), b""]:
        add("pairing", m.bls12_pairing, inp)
    for u in [0, 1, 0x156c8a6a2c184569d69a76be144b5cdc5141d2d2ca4fe341f011e25e3969c55ad9e9b9ce2eb833c81a908e5fa4ac5f03, rng.scalar256() % m.P]:
        add("map_fp_to_g1", h2c.bls12_map_fp_to_g1, m.encode_fp(u))
    add("map_fp_to_g1", h2c.bls12_map_fp_to_g1, bytes(63))
    add("map_fp_to_g1", h2c.bls12_map_fp_to_g1, bytes(16) + m.P.to_bytes(48, "big"))
    for u in [(0, 0), (1, 2), (rng.scalar256() % m.P, rng.scalar256() % m.P)]:
        add("map_fp2_to_g2", h2c.bls12_map_fp2_to_g2, m.encode_fp(u[0]) + m.encode_fp(u[1]))
    add("map_fp2_to_g2", h2c.bls12_map_fp2_to_g2, bytes(129))
    return v


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "kat.json"), "w") as f:
        json.dump(kat(), f, indent=0)
    print("kat.json written")
    if "--kat-only" in sys.argv:
        sys.exit(0)
    # 2^21..2^23: the totals of bench.py's weak-scaling runs on 2 / 4 / 8 GPUs
    for wl, l2 in [("g1msm", 10), ("g1msm", 16), ("g1msm", 20), ("g1msm", 21), ("g1msm", 22), ("g1msm", 23),
                   ("g2msm", 10), ("g2msm", 16)]:
        with open(os.path.join(OUT, "%s_2p%d.hex" % (wl, l2)), "w") as f:
            f.write(analytic(wl, l2).hex() + "\n")
        print(wl, l2, "written")
