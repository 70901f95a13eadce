#!/usr/bin/env python3
"""Constructible EIP-2537 vector families, written in the reference's OWN two vector formats and file
names (the files its build.sh:13-52 downloads and its C / Go / Rust harnesses read: CSV src/test.c:63-72,
JSON go/blst_eip2537_test.go:18-29), into tests/golden/eip2537_constructed/.

These are NOT the published files (absent offline).  Every expected value is constructed, not recalled:
inputs are known multiples of the group generators (so a sum, a product or a pairing identity has a
closed form: e([a]G1,[b]G2) * e([-ab]G1, G2) = 1), the outputs come from the independent big-integer
Python model (oracle/pymodel), and the failure files carry the error class the reference's C harness
requires for that file (src/test.c:144-165 g1_not_on_curve -> bls12_g1mul == POINT_NOT_ON_CURVE;
:337-358 g2_not_on_curve -> bls12_g2mul; :481-511 invalid_subgroup_for_pairing -> POINT_NOT_IN_SUBGROUP;
:564-585 / :638-659 invalid_fp(2)_encoding -> map == INVALID_ELEMENT).  tools/run_kat.py ingests them
exactly as it would ingest the published files.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pymodel"))
import bls12_381 as m  # noqa: E402
import h2c  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "eip2537_constructed")
e1, e2, es, efp = m.encode_g1, m.encode_g2, m.encode_scalar, m.encode_fp
G1, G2, R, P = m.G1, m.G2, m.R, m.P


def build():
    rng = m.SplitMix64(0xE1B2537)
    mult1 = lambda k: m.g1_mul(G1, k % R)
    mult2 = lambda k: m.g2_mul(G2, k % R)
    ks = [1, 2, 3, 0xabcdef, R - 1, rng.scalar256() % R, rng.scalar256() % R]
    files = {}

    def ok(fn, inp):
        code, out = m.call(fn, inp)
        assert code == 0, (fn.__name__, code)
        return inp, out

    # ---- add / mul: closed forms on generator multiples  [a]G + [b]G = [a+b]G,  [k][a]G = [ka]G
    g1add = [ok(m.bls12_g1add, e1(mult1(a)) + e1(mult1(b))) for a, b in zip(ks, ks[1:] + ks[:1])]
    g1add += [ok(m.bls12_g1add, e1(mult1(5)) + e1(mult1(R - 5))), ok(m.bls12_g1add, e1(None) + e1(mult1(7))),
              ok(m.bls12_g1add, e1(mult1(9)) + e1(mult1(9))), ok(m.bls12_g1add, bytes(256))]
    for (inp, out), (a, b) in zip(g1add, zip(ks, ks[1:] + ks[:1])):
        assert out == e1(mult1(a + b))
    g2add = [ok(m.bls12_g2add, e2(mult2(a)) + e2(mult2(b))) for a, b in zip(ks[:5], ks[1:6])]
    g2add += [ok(m.bls12_g2add, e2(mult2(5)) + e2(mult2(R - 5))), ok(m.bls12_g2add, e2(None) + e2(mult2(7))), ok(m.bls12_g2add, bytes(512))]
    scal = [0, 1, 2, R - 1, R, R + 1, 2 ** 255, 2 ** 256 - 1, rng.scalar256()]
    g1mul = [ok(m.bls12_g1mul, e1(mult1(0xabc)) + es(k)) for k in scal] + [ok(m.bls12_g1mul, e1(None) + es(77))]
    for (inp, out), k in zip(g1mul, scal):
        assert out == e1(mult1(0xabc * k))
    g2mul = [ok(m.bls12_g2mul, e2(mult2(0xdef)) + es(k)) for k in scal[:7]] + [ok(m.bls12_g2mul, e2(None) + es(77))]
    # ---- multiexp: sum k_i [a_i]G = [sum k_i a_i]G; the C harness runs every row through _bc too (:208-228)
    g1msm, g2msm = [], []
    for n in (1, 2, 3, 4, 5, 8, 16, 33):
        pairs = [((rng.scalar256() % R), rng.scalar256()) for _ in range(n)]
        inp = b"".join(e1(mult1(a)) + es(k) for a, k in pairs)
        row = ok(m.bls12_g1multiexp, inp)
        assert row[1] == e1(mult1(sum(a * k for a, k in pairs)))
        g1msm.append(row)
    g1msm.append(ok(m.bls12_g1multiexp, e1(mult1(3)) + es(5) + e1(None) + es(9) + e1(mult1(R - 3)) + es(5)))      # -> infinity
    for n in (1, 2, 4, 5, 9):
        pairs = [((rng.scalar256() % R), rng.scalar256()) for _ in range(n)]
        inp = b"".join(e2(mult2(a)) + es(k) for a, k in pairs)
        row = ok(m.bls12_g2multiexp, inp)
        assert row[1] == e2(mult2(sum(a * k for a, k in pairs)))
        g2msm.append(row)
    # ---- pairing identities from generator multiples
    one, zero = bytes(31) + b"\x01", bytes(32)
    pr = []

    def pairs_bytes(ps):
        return b"".join(e1(p) + e2(q) for p, q in ps)
    a, b, c, d = 0x1234567, 0x89abcdef, rng.scalar256() % R, rng.scalar256() % R
    pr.append((pairs_bytes([(mult1(a), mult2(b)), (mult1(-a * b), G2)]), one))
    pr.append((pairs_bytes([(mult1(a), mult2(b)), (mult1(1 - a * b), G2)]), zero))
    pr.append((pairs_bytes([(mult1(c), G2), (G1, mult2(-c))]), one))                       # bilinearity
    pr.append((pairs_bytes([(mult1(c), mult2(d)), (mult1(d), mult2(-c))]), one))
    pr.append((pairs_bytes([(G1, G2)]), zero))                                              # non-degenerate
    pr.append((pairs_bytes([(None, G2)]), one))
    pr.append((pairs_bytes([(G1, None)]), one))
    pr.append((pairs_bytes([(None, None), (mult1(a), mult2(b)), (mult1(-a * b), G2)]), one))
    pr.append((pairs_bytes([(mult1(2), mult2(3)), (mult1(5), mult2(7)), (mult1(-41), G2)]), one))
    pr.append((pairs_bytes([(mult1(2), mult2(3)), (mult1(5), mult2(7)), (mult1(-40), G2)]), zero))
    for inp, want in pr:
        assert m.call(m.bls12_pairing, inp) == (0, want)
    # ---- map to curve (model reproduces the RFC 9380 appendix J vectors: tests/test_h2c.py)
    fp1 = [ok(h2c.bls12_map_fp_to_g1, efp(u)) for u in (0, 1, 2, P - 1, rng.scalar256() * rng.scalar256() % P)]
    fp2 = [ok(h2c.bls12_map_fp2_to_g2, efp(u0) + efp(u1)) for u0, u1 in ((0, 0), (1, 0), (0, 1), (P - 1, 7), (rng.scalar256(), rng.scalar256()))]
    # ---- failure classes
    ns1, ns2 = m.random_g1(rng, False), m.random_g2(rng, False)
    assert not m.g1_in_subgroup(ns1) and not m.g2_in_subgroup(ns2)
    off1 = [efp(1) + efp(1), efp(2) + efp(2), efp(0) + efp(1), e1(G1)[:64] + efp(5)]
    off2 = [efp(1) * 4, efp(0) * 3 + efp(1), e2(G2)[:128] + efp(5) + efp(6)]
    g1_not_on_curve = [pt + es(3) for pt in off1]                                           # 160-byte g1mul inputs
    g2_not_on_curve = [pt + es(3) for pt in off2]                                           # 288-byte g2mul inputs
    bad_sub = [pairs_bytes([(ns1, G2)]), pairs_bytes([(G1, ns2)]), pairs_bytes([(G1, G2), (ns1, G2)]),
               pairs_bytes([(mult1(3), mult2(4)), (G1, ns2)]), pairs_bytes([((0, 2), G2)])]
    bad_fp = [bytes(16) + P.to_bytes(48, "big"), bytes(16) + (P + 1).to_bytes(48, "big"), bytes(16) + b"\xff" * 48,
              b"\x01" + bytes(15) + (5).to_bytes(48, "big"), bytes(15) + b"\x01" + (5).to_bytes(48, "big")]
    bad_fp2 = [efp(1) + b for b in bad_fp] + [bad_fp[0] + efp(1)]
    for inp in g1_not_on_curve:
        assert m.call(m.bls12_g1mul, inp)[0] == 1
    for inp in g2_not_on_curve:
        assert m.call(m.bls12_g2mul, inp)[0] == 1
    for inp in bad_sub:
        assert m.call(m.bls12_pairing, inp)[0] == 2
    for inp in bad_fp:
        assert m.call(h2c.bls12_map_fp_to_g1, inp)[0] == 3
    for inp in bad_fp2:
        assert m.call(h2c.bls12_map_fp2_to_g2, inp)[0] == 3

    files["g1_add.csv"], files["g1_mul.csv"], files["g1_multiexp.csv"] = g1add, g1mul, g1msm
    files["g2_add.csv"], files["g2_mul.csv"], files["g2_multiexp.csv"] = g2add, g2mul, g2msm
    files["pairing.csv"], files["fp_to_g1.csv"], files["fp2_to_g2.csv"] = pr, fp1, fp2
    files["g1_not_on_curve.csv"] = [(i, b"") for i in g1_not_on_curve]
    files["g2_not_on_curve.csv"] = [(i, b"") for i in g2_not_on_curve]
    files["invalid_subgroup_for_pairing.csv"] = [(i, b"") for i in bad_sub]
    files["invalid_fp_encoding.csv"] = [(i, b"") for i in bad_fp]
    files["invalid_fp2_encoding.csv"] = [(i, b"") for i in bad_fp2]
    js = {"blsG1Add.json": g1add, "blsG1Mul.json": g1mul, "blsG1MultiExp.json": g1msm, "blsG2Add.json": g2add,
          "blsG2Mul.json": g2mul, "blsG2MultiExp.json": g2msm, "blsPairing.json": pr, "blsMapG1.json": fp1, "blsMapG2.json": fp2}
    fail = {"fail-blsG1Add.json": [off1[0] + e1(G1), e1(G1) + off1[1], bytes(255), bad_fp[0] + efp(1) + e1(G1)],
            "fail-blsG1Mul.json": g1_not_on_curve + [bytes(159)],
            "fail-blsG1MultiExp.json": [g1msm[3][0][:-160] + g1_not_on_curve[0], b"", g1msm[2][0] + b"\x00"],
            "fail-blsG2Add.json": [off2[0] + e2(G2), bytes(511)],
            "fail-blsG2Mul.json": g2_not_on_curve + [bytes(287)],
            "fail-blsG2MultiExp.json": [g2msm[1][0][:-288] + g2_not_on_curve[0], b""],
            "fail-blsMapG1.json": bad_fp + [bytes(63)], "fail-blsMapG2.json": bad_fp2 + [bytes(127)],
            "fail-blsPairing.json": bad_sub + [b"", bytes(383), pairs_bytes([(G1, G2)])[:-256] + off2[0]]}
    return files, js, fail


def main():
    files, js, fail = build()
    os.makedirs(OUT, exist_ok=True)
    for name, rows in files.items():
        with open(os.path.join(OUT, name), "w", newline="") as f:
            w = csv.writer(f, lineterminator="\n")
            w.writerow(["input", "result"])
            for inp, out in rows:
                w.writerow([inp.hex(), out.hex()])
    for name, rows in js.items():
        with open(os.path.join(OUT, name), "w") as f:
            json.dump([{"Input": i.hex(), "Expected": o.hex(), "Name": "%s_%d" % (name[:-5], n), "Gas": 0, "NoBenchmark": True}
                       for n, (i, o) in enumerate(rows)], f, indent=0)
    for name, rows in fail.items():
        with open(os.path.join(OUT, name), "w") as f:
            json.dump([{"Input": i.hex(), "ExpectedError": "constructed failure class", "Name": "%s_%d" % (name[:-5], n)}
                       for n, i in enumerate(rows)], f, indent=0)
    n = sum(len(r) for r in files.values()) + sum(len(r) for r in js.values()) + sum(len(r) for r in fail.values())
    print("wrote %d vectors in %d files to %s" % (n, len(files) + len(js) + len(fail), OUT))


if __name__ == "__main__":
    main()
