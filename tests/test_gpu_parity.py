"""GPU parity: the HIP path through the C-ABI vs the CPU oracle on the same seeded inputs
(bit-exact: this is integer work).  Covers the dispatcher thresholds of the reference
(src/eip2537.c:550-560), wave boundaries, adversarial inputs and the error order."""
import os

import pytest

import bls12_381 as m
from conftest import call_x

pytestmark = pytest.mark.gpu

A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321


@pytest.mark.parametrize("n", [1, 2, 4, 5, 63, 64, 65, 257, 1000, 4096])
def test_g1_msm_sizes(X, clib, n):
    inp = clib.gen_msm_input("g1", n, A, B, 0x25370000 + n)
    assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp)


@pytest.mark.parametrize("n", [1, 2, 4, 5, 64, 65, 300, 1024])
def test_g2_msm_sizes(X, clib, n):
    inp = clib.gen_msm_input("g2", n, A, B, 0x25370100 + n)
    assert call_x(X.g2_multiexp, inp) == clib.call("bls12_g2multiexp", inp)


@pytest.mark.parametrize("c", [4, 7, 8, 11, 13, 16])
def test_g1_msm_window_widths(X, clib, c):
    inp = clib.gen_msm_input("g1", 777, A, B, 99)
    want = clib.call("bls12_g1multiexp", inp)
    X.set_window(c)
    try:
        assert call_x(X.g1_multiexp, inp) == want
    finally:
        X.set_window(0)


def _rec1(pt, k):
    return m.encode_g1(pt) + m.encode_scalar(k)


def test_g1_msm_adversarial(X, clib):
    rng = m.SplitMix64(5)
    P = m.g1_mul(m.G1, 0xabcdef)
    Q = m.g1_mul(m.G1, 0x123457)
    nP = m.ec_neg(m.FP, P)
    ks = [0, 1, m.R - 1, m.R, m.R + 1, 2 ** 255, 2 ** 256 - 1]
    cases = {
        "dup_same_scalar": b"".join(_rec1(P, 7) for _ in range(9)),
        "p_and_minus_p": b"".join(_rec1(P if i % 2 else nP, 12345) for i in range(10)),
        "all_equal_scalars": b"".join(_rec1(m.g1_mul(m.G1, i + 1), 2 ** 256 - 1) for i in range(40)),
        "special_scalars": b"".join(_rec1(Q, k) for k in ks),
        "infinity_inside": _rec1(P, 5) + _rec1(None, 77) + _rec1(Q, 9) + _rec1(None, 0) + _rec1(P, 1),
        "all_infinity": b"".join(_rec1(None, rng.scalar256()) for _ in range(6)),
        "all_zero_scalars": b"".join(_rec1(m.g1_mul(m.G1, i + 3), 0) for i in range(6)),
        "order3_point": b"".join(_rec1((0, 2), k) for k in [1, 2, 4, 5, 2 ** 256 - 1, 7]),
        "non_subgroup": b"".join(_rec1(m.random_g1(rng, False), rng.scalar256()) for _ in range(7)),
        "sum_to_infinity": _rec1(P, 5) + _rec1(nP, 5) + _rec1(Q, m.R) + _rec1(P, 0) + _rec1(None, 3),
    }
    for name, inp in cases.items():
        assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp), name
        assert call_x(X.g1_multiexp_bc, inp) == clib.call("bls12_g1multiexp_bc", inp), name
        assert call_x(X.g1_multiexp_naive, inp) == clib.call("bls12_g1multiexp_naive", inp), name
    # the same inputs through the plan of the large sizes (c = 16: one lane per task, limb-form accumulate,
    # csrc/limb30.h) -- equal, opposite and repeated entries meet its doubling and cancellation paths
    X.set_window(16)
    try:
        for name, inp in cases.items():
            assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp), name + " (c = 16)"
        assert X.last_plan()["kernel"] == ("k_msm_accum<eip::Fp>" if os.environ.get("EIP2537_LIMB_FORM") == "0" else "k_msm_accum_l")
    finally:
        X.set_window(0)


def test_g2_msm_adversarial(X, clib):
    rng = m.SplitMix64(6)
    P = m.g2_mul(m.G2, 0xabcdef)
    nP = m.ec_neg(m.FP2, P)
    rec = lambda pt, k: m.encode_g2(pt) + m.encode_scalar(k)
    cases = {
        "dup": b"".join(rec(P, 7) for _ in range(6)),
        "p_minus_p": b"".join(rec(P if i % 2 else nP, 999) for i in range(6)),
        "specials": b"".join(rec(P, k) for k in [0, 1, m.R, m.R + 1, 2 ** 256 - 1, 3]),
        "inf_inside": rec(P, 5) + rec(None, 7) + rec(nP, 4) + rec(None, 0) + rec(P, 2),
        "non_subgroup": b"".join(rec(m.random_g2(rng, False), rng.scalar256()) for _ in range(5)),
    }
    for name, inp in cases.items():
        assert call_x(X.g2_multiexp, inp) == clib.call("bls12_g2multiexp", inp), name


def test_msm_error_order(X, clib):
    inp = bytearray(clib.gen_msm_input("g1", 200, A, B, 1))
    # record 150: off-curve (1,1); record 37: pad byte set -> INVALID_ELEMENT of record 37 wins
    inp[150 * 160:150 * 160 + 128] = m.encode_fp(1) + m.encode_fp(1)
    assert call_x(X.g1_multiexp, bytes(inp)) == (1, None)
    inp[37 * 160 + 3] = 1
    assert call_x(X.g1_multiexp, bytes(inp)) == (3, None)
    assert clib.call("bls12_g1multiexp", bytes(inp)) == (3, None)
    # x = p is invalid; both coordinates decoded before the verdict
    inp2 = bytearray(clib.gen_msm_input("g1", 10, A, B, 1))
    inp2[0:128] = m.encode_fp(1) + bytes(16) + m.P.to_bytes(48, "big")
    assert call_x(X.g1_multiexp, bytes(inp2)) == (3, None)
    assert call_x(X.g1_multiexp, b"") == (5, None)
    assert call_x(X.g1_multiexp, bytes(161)) == (5, None)
    assert call_x(X.g2_multiexp, bytes(287)) == (5, None)


def _pairs(ps):
    return b"".join(m.encode_g1(p) + m.encode_g2(q) for p, q in ps)


def test_pairing_small(X, clib):
    G1, G2 = m.G1, m.G2
    neg = lambda p: m.ec_neg(m.FP, p)
    cases = [
        _pairs([(m.g1_mul(G1, 3), G2), (neg(G1), m.g2_mul(G2, 3))]),
        _pairs([(m.g1_mul(G1, 3), G2), (neg(G1), m.g2_mul(G2, 4))]),
        _pairs([(G1, G2)]),
        _pairs([(None, G2), (G1, None)]),
        _pairs([(None, None)]),
        _pairs([(m.g1_mul(G1, 6), m.g2_mul(G2, 35)), (neg(m.g1_mul(G1, 10)), m.g2_mul(G2, 21)), (None, G2)]),
    ]
    for i, inp in enumerate(cases):
        assert call_x(X.pairing, inp) == clib.call("bls12_pairing", inp), i
    assert call_x(X.pairing, cases[0])[1][-1] == 1
    assert call_x(X.pairing, cases[1])[1][-1] == 0


@pytest.mark.parametrize("k", [63, 64, 65, 200])
def test_pairing_batches(X, clib, k):
    a0, a1, b0, b1 = A, B, B ^ 0x55, A ^ 0x33
    base = bytearray(clib.gen_pairing_input(k, a0, a1, b0, b1))
    # fix up the last pair so that the product is one: e([c]G1, G2) with c = -sum a_i b_i
    s = sum(((a0 + i * a1) % m.R) * ((b0 + i * b1) % m.R) for i in range(k - 1)) % m.R
    good = bytes(base[:(k - 1) * 384]) + m.encode_g1(m.g1_mul(m.G1, (-s) % m.R)) + m.encode_g2(m.G2)
    bad = bytes(base[:(k - 1) * 384]) + m.encode_g1(m.g1_mul(m.G1, (1 - s) % m.R)) + m.encode_g2(m.G2)
    assert call_x(X.pairing, good) == (0, bytes(31) + b"\x01")
    assert call_x(X.pairing, bad) == (0, bytes(32))
    assert clib.call("bls12_pairing", good) == (0, bytes(31) + b"\x01")


def test_pairing_error_order(X, clib):
    rng = m.SplitMix64(8)
    k = 20
    base = bytearray(clib.gen_pairing_input(k, A, B, B, A))
    ns1 = m.random_g1(rng, False)
    ns2 = m.random_g2(rng, False)
    t = bytearray(base)
    t[7 * 384:7 * 384 + 128] = m.encode_g1(ns1)                      # pair 7: G1 not in subgroup (2)
    t[3 * 384 + 128:3 * 384 + 384] = m.encode_fp(1) * 4              # pair 3: G2 off curve (1)
    assert call_x(X.pairing, bytes(t)) == (1, None)
    assert clib.call("bls12_pairing", bytes(t)) == (1, None)
    t = bytearray(base)
    t[5 * 384 + 128:5 * 384 + 384] = m.encode_g2(ns2)               # same pair: G2 not in subgroup
    t[5 * 384:5 * 384 + 128] = m.encode_fp(1) + m.encode_fp(1)       # and G1 off curve -> G1 first
    assert call_x(X.pairing, bytes(t)) == (1, None)
    t[5 * 384:5 * 384 + 128] = m.encode_g1(ns1)                      # G1 not in subgroup beats G2
    assert call_x(X.pairing, bytes(t)) == (2, None)
    t = bytearray(base)
    t[9 * 384 + 128:9 * 384 + 384] = m.encode_g2(ns2)
    assert call_x(X.pairing, bytes(t)) == (2, None)
    assert call_x(X.pairing, b"") == (5, None)
    assert call_x(X.pairing, bytes(383)) == (5, None)


def test_concurrent_callers(X, clib):
    """The ABI is stateless and callers may be concurrent (cargo test threads, goroutines --
    SURVEY.md 8b): 27 threads -- more than the library has engine slots -- hammer different
    precompiles, one of them with a bad record whose error must stay on its own call."""
    import threading
    g1 = [clib.gen_msm_input("g1", 100 + 37 * t, A, B, 1000 + t) for t in range(4)]
    g2 = [clib.gen_msm_input("g2", 40 + 11 * t, A, B, 2000 + t) for t in range(2)]
    pr = [_pairs([(m.g1_mul(m.G1, 3 + t), m.G2), (m.ec_neg(m.FP, m.G1), m.g2_mul(m.G2, 3 + t))]) for t in range(2)]
    want = [clib.call("bls12_g1multiexp", b) for b in g1] + [clib.call("bls12_g2multiexp", b) for b in g2] + \
           [clib.call("bls12_pairing", b) for b in pr]
    jobs = [(X.g1_multiexp, b) for b in g1] + [(X.g2_multiexp, b) for b in g2] + [(X.pairing, b) for b in pr]
    broken = bytearray(g1[1])
    broken[160 * 57 + 20] ^= 0x40                      # x of record 57 leaves the curve
    jobs.append((X.g1_multiexp, bytes(broken)))
    want.append(clib.call("bls12_g1multiexp", bytes(broken)))
    assert want[-1][0] != 0
    jobs = jobs * 3
    want = want * 3
    bad = []

    def work(i):
        fn, inp = jobs[i]
        for _ in range(6):
            if call_x(fn, inp) != want[i]:
                bad.append(i)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad


def test_forced_window_width_without_a_valid_plan_falls_back(X, clib):
    """eip2537_hip_set_window is a bench / sweep knob; c = 14 has no valid plan (its merged top
    window would need 18 bits) and used to leave an empty plan (SIGFPE in the host code)."""
    inp = clib.gen_msm_input("g1", 300, A, B, 77)
    want = clib.call("bls12_g1multiexp", inp)
    try:
        for c in (14, 4, 16, 0):
            X.set_window(c)
            assert call_x(X.g1_multiexp, inp) == want, c
    finally:
        X.set_window(0)


def test_pairing_small_order_g1_points(X, clib):
    """The G1 membership chain [z^2]P meets P, -P and infinity on the way when P has small order
    (the complete-addition branches of the 4-lane kernel): (0, 2) has order 3."""
    p3 = m.encode_g1((0, 2))
    g2 = m.encode_g2(m.G2)
    good = m.encode_g1(m.G1) + g2
    for inp in (p3 + g2, good + p3 + g2, p3 + g2 + good, good * 5 + p3 + g2 + good * 3):
        want = clib.call("bls12_pairing", inp)
        assert want == (2, None)
        assert call_x(X.pairing, inp) == want
    # (0, -2) as well, and an order-3 point beside the point at infinity
    n3 = m.encode_g1((0, m.P - 2))
    assert call_x(X.pairing, n3 + g2) == clib.call("bls12_pairing", n3 + g2) == (2, None)
    inp = bytes(128) + g2 + p3 + g2
    assert call_x(X.pairing, inp) == clib.call("bls12_pairing", inp) == (2, None)


def test_device_field_products_selftest(X):
    """The column products the kernels use (radix 2^30; canonical and lazy, product and square) against
    the independent 12 x 32-bit product, on the device: guards against the code-generation problem
    recorded in field.h (the host build of the same source was correct, the device build was not)."""
    for seed in (1, 0x2537):
        assert X.field_selftest(seed, 1 << 18) == (0, 0, 0, 0)


def test_device_limb_form_selftest(X):
    """The limb-form primitives of the G1 MSM and the pairing kernels (13 x 30-bit limbs, Montgomery factor 2^390:
    mulL, sqrL, mul2L, fp_mul2_cols30, conversions, weak reduction, linear steps, zero test) on the DEVICE against the
    12 x 32-bit product, with operands grown to the kernels' bounds -- the host tools run the same source on the host only."""
    for seed in (1, 0x2537):
        assert X.limb_selftest(seed, 1 << 17) == (0,) * 8
