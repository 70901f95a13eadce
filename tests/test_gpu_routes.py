"""The small-call crossover: below a measured size the reference-ABI multiexp / pairing
calls run the library's own host code, above it the GPU.  Both routes must give the oracle's bytes and
the reference's error order (src/eip2537.c:550-560, 1036-1053), so every case here runs with the route
pinned to the GPU (0), to the host code (1) and with the default rule (-1)."""
import pytest

import bls12_381 as m
from conftest import call_x

pytestmark = pytest.mark.gpu

A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321


@pytest.fixture(params=[0, 1, -1], ids=["gpu-route", "host-route", "default-route"])
def route(request, X):
    X.set_route(request.param)
    yield request.param
    X.set_route(-1)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 64])
def test_g1_msm_small_both_routes(X, clib, route, n):
    inp = clib.gen_msm_input("g1", n, A, B, 0x77 + n)
    want = clib.call("bls12_g1multiexp", inp)
    assert call_x(X.g1_multiexp, inp) == want
    assert call_x(X.g1_multiexp_naive, inp) == want
    assert call_x(X.g1_multiexp_bc, inp) == want


@pytest.mark.parametrize("n", [1, 2, 3, 5, 33])
def test_g2_msm_small_both_routes(X, clib, route, n):
    inp = clib.gen_msm_input("g2", n, A, B, 0x99 + n)
    assert call_x(X.g2_multiexp, inp) == clib.call("bls12_g2multiexp", inp)


def _pairs(ps):
    return b"".join(m.encode_g1(p) + m.encode_g2(q) for p, q in ps)


def test_pairing_small_both_routes(X, clib, route):
    G1, G2 = m.G1, m.G2
    neg = lambda p: m.ec_neg(m.FP, p)
    cases = [
        _pairs([(G1, G2)]),
        _pairs([(None, None)]),
        _pairs([(None, G2)]),
        _pairs([(G1, None)]),
        _pairs([(m.g1_mul(G1, 3), G2), (neg(G1), m.g2_mul(G2, 3))]),            # == 1
        _pairs([(m.g1_mul(G1, 3), G2), (neg(G1), m.g2_mul(G2, 4))]),            # != 1
        _pairs([(None, G2), (G1, None)]),
        _pairs([(m.g1_mul(G1, 6), m.g2_mul(G2, 35)), (neg(m.g1_mul(G1, 10)), m.g2_mul(G2, 21)), (None, G2)]),
        _pairs([(m.g1_mul(G1, i + 2), m.g2_mul(G2, 5 * i + 1)) for i in range(8)]),
    ]
    for i, inp in enumerate(cases):
        assert call_x(X.pairing, inp) == clib.call("bls12_pairing", inp), i
    assert call_x(X.pairing, cases[4])[1][-1] == 1 and call_x(X.pairing, cases[5])[1][-1] == 0


def test_adversarial_and_error_order_both_routes(X, clib, route):
    rng = m.SplitMix64(11)
    P = m.g1_mul(m.G1, 0xabcdef)
    rec = lambda pt, k: m.encode_g1(pt) + m.encode_scalar(k)
    for inp in [rec(P, 0), rec(None, 5), rec(P, m.R), rec(P, 2 ** 256 - 1), rec((0, 2), 5),
                rec(m.random_g1(rng, False), rng.scalar256()), rec(P, 7) + rec(m.ec_neg(m.FP, P), 7)]:
        assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp)
    # MSM: first bad record in input order
    bad = bytearray(clib.gen_msm_input("g1", 6, A, B, 1))
    bad[4 * 160:4 * 160 + 128] = m.encode_fp(1) + m.encode_fp(1)              # record 4 off curve (1)
    assert call_x(X.g1_multiexp, bytes(bad)) == (1, None) == clib.call("bls12_g1multiexp", bytes(bad))
    bad[1 * 160 + 3] = 1                                                       # record 1 pad byte (3) wins
    assert call_x(X.g1_multiexp, bytes(bad)) == (3, None) == clib.call("bls12_g1multiexp", bytes(bad))
    one = bytearray(rec(P, 3))
    one[0:128] = m.encode_fp(1) + bytes(16) + m.P.to_bytes(48, "big")         # y = p: invalid element
    assert call_x(X.g1_multiexp, bytes(one)) == (3, None)
    g2bad = bytearray(clib.gen_msm_input("g2", 2, A, B, 1))
    g2bad[288:288 + 256] = m.encode_fp(1) * 4
    assert call_x(X.g2_multiexp, bytes(g2bad)) == (1, None) == clib.call("bls12_g2multiexp", bytes(g2bad))
    # pairing: inside a pair G1 decode -> G1 subgroup -> G2 decode -> G2 subgroup; lowest pair first
    ns1, ns2 = m.random_g1(rng, False), m.random_g2(rng, False)
    for k in (1, 2, 5):
        base = bytearray(clib.gen_pairing_input(k, A, B, B, A))
        last = (k - 1) * 384
        t = bytearray(base)
        t[last + 128:last + 384] = m.encode_g2(ns2)
        assert call_x(X.pairing, bytes(t)) == (2, None) == clib.call("bls12_pairing", bytes(t))
        t[last:last + 128] = m.encode_fp(1) + m.encode_fp(1)                  # G1 off curve beats G2 subgroup
        assert call_x(X.pairing, bytes(t)) == (1, None) == clib.call("bls12_pairing", bytes(t))
        t[last:last + 128] = m.encode_g1(ns1)                                 # G1 subgroup beats G2 subgroup
        assert call_x(X.pairing, bytes(t)) == (2, None)
        t = bytearray(base)
        t[last + 128:last + 384] = m.encode_fp(1) * 4                         # G2 off curve
        assert call_x(X.pairing, bytes(t)) == (1, None)
        if k > 1:
            t[0:128] = m.encode_g1(ns1)                                       # pair 0 beats the last pair
            assert call_x(X.pairing, bytes(t)) == (2, None) == clib.call("bls12_pairing", bytes(t))
    assert call_x(X.pairing, b"") == (5, None) and call_x(X.g1_multiexp, bytes(159)) == (5, None)
