"""map_fp_to_g1 / map_fp2_to_g2.  Unlike the rest of the ABI this part IS
pinned by external ground truth: the RFC 9380 appendix-J vectors (the standard blst's
map_to_g1/g2 implements; EIP-2537's own map vectors use the NU inputs).  The isogeny tables are
derived from scratch (tools/derive_isogeny.py); model, C oracle and the product's host code must
all reproduce the vectors and agree with each other.  CPU only: the map precompiles are host code."""
import json
import os

import bls12_381 as m
import h2c
from conftest import ROOT, call_x

with open(os.path.join(ROOT, "tests", "golden", "rfc9380_vectors.json")) as f:
    V = json.load(f)
H = lambda s: int(s, 16)


def test_model_reproduces_rfc9380_vectors():
    v = V["g1_ro"]
    pt, u = h2c.hash_to_curve_g1(b"", v["dst"].encode())
    assert (u[0], u[1]) == (H(v["u0"]), H(v["u1"])) and pt == (H(v["px"]), H(v["py"]))
    v = V["g1_nu"]
    u = h2c.hash_to_field(b"", v["dst"].encode(), 1, 1)[0]
    assert u == H(v["u0"]) and h2c.map_fp_to_g1(u) == (H(v["px"]), H(v["py"]))
    v = V["g2_ro"]
    pt, _ = h2c.hash_to_curve_g2(b"", v["dst"].encode())
    assert pt == (tuple(map(H, v["px"])), tuple(map(H, v["py"])))
    v = V["g2_nu"]
    u = h2c.hash_to_field(b"", v["dst"].encode(), 1, 2)[0]
    assert u == tuple(map(H, v["u0"])) and h2c.map_fp2_to_g2(u) == (tuple(map(H, v["px"])), tuple(map(H, v["py"])))


def test_isogeny_tables_are_isogenies():
    """the derived maps send E' to the target curve, kernel points excepted, and are homomorphisms"""
    rng = m.SplitMix64(3)
    pts = []
    while len(pts) < 3:
        x = rng.scalar256() % m.P
        y = m.fp_sqrt((x * x * x + h2c.G1_A * x + h2c.G1_B) % m.P)
        if y is not None:
            pts.append((x, y))
    for pt in pts:
        assert m.ec_on_curve(m.FP, m.B1, h2c.iso_map(m.FP, "g1", pt))
    # E1' has the order of E (isogenous): [h1 r]P' = infinity on E1'
    def add_e1p(p1, p2):     # affine add on y^2 = x^3 + A x + B
        if p1 is None: return p2
        if p2 is None: return p1
        (x1, y1), (x2, y2) = p1, p2
        if x1 == x2:
            if (y1 + y2) % m.P == 0: return None
            lam = (3 * x1 * x1 + h2c.G1_A) * pow(2 * y1, -1, m.P) % m.P
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, m.P) % m.P
        x3 = (lam * lam - x1 - x2) % m.P
        return (x3, (lam * (x1 - x3) - y1) % m.P)
    acc, k, base = None, m.H1 * m.R, pts[0]
    while k:
        if k & 1: acc = add_e1p(acc, base)
        base = add_e1p(base, base)
        k >>= 1
    assert acc is None
    s = add_e1p(pts[0], pts[1])
    assert h2c.iso_map(m.FP, "g1", s) == m.g1_add(h2c.iso_map(m.FP, "g1", pts[0]), h2c.iso_map(m.FP, "g1", pts[1]))


def test_oracle_and_product_match_model_and_vectors(X, clib):
    v = V["g1_nu"]
    inp = m.encode_fp(H(v["u0"]))
    want = (0, m.encode_fp(H(v["px"])) + m.encode_fp(H(v["py"])))
    assert clib.call("bls12_map_fp_to_g1", inp) == want and call_x(X.map_fp_to_g1, inp) == want
    v = V["g2_nu"]
    inp = b"".join(m.encode_fp(H(c)) for c in v["u0"])
    want = (0, b"".join(m.encode_fp(H(c)) for c in v["px"] + v["py"]))
    assert clib.call("bls12_map_fp2_to_g2", inp) == want and call_x(X.map_fp2_to_g2, inp) == want
    rng = m.SplitMix64(8)
    for i in range(12):
        u = [0, 1, m.P - 1][i] if i < 3 else rng.scalar256() * rng.scalar256() % m.P
        inp = m.encode_fp(u)
        want = (0, m.encode_g1(h2c.map_fp_to_g1(u)))
        assert clib.call("bls12_map_fp_to_g1", inp) == want and call_x(X.map_fp_to_g1, inp) == want, i
    for i in range(6):
        u = [(0, 0), (1, 0), (0, 1)][i] if i < 3 else (rng.scalar256() * rng.scalar256() % m.P, rng.scalar256() * rng.scalar256() % m.P)
        inp = m.encode_fp(u[0]) + m.encode_fp(u[1])
        want = (0, m.encode_g2(h2c.map_fp2_to_g2(u)))
        assert clib.call("bls12_map_fp2_to_g2", inp) == want and call_x(X.map_fp2_to_g2, inp) == want, i
    # outputs land in the prime-order subgroups; bad inputs give the reference's codes
    assert clib.in_subgroup("g1", X.map_fp_to_g1(m.encode_fp(12345))) == 1
    assert clib.in_subgroup("g2", X.map_fp2_to_g2(m.encode_fp(3) + m.encode_fp(9))) == 1
    for fn, name, n in [(X.map_fp_to_g1, "bls12_map_fp_to_g1", 64), (X.map_fp2_to_g2, "bls12_map_fp2_to_g2", 128)]:
        assert call_x(fn, bytes(n - 1)) == clib.call(name, bytes(n - 1)) == (5, None)
        bad = bytes(n - 48) + m.P.to_bytes(48, "big")
        assert call_x(fn, bad) == clib.call(name, bad) == (3, None)
        bad = b"\x01" + bytes(n - 1)
        assert call_x(fn, bad) == clib.call(name, bad) == (3, None)
