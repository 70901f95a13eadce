"""Seeded randomized parity: random batch sizes, mixtures of ordinary / infinity / duplicate /
negated / non-subgroup points, special scalars, and random single-record corruptions; the HIP path
through the C-ABI must return exactly what the CPU oracle returns (output bytes or error code)."""
import random

import pytest

import bls12_381 as m
from conftest import call_x

pytestmark = pytest.mark.gpu

SPECIAL_K = [0, 1, 2, m.R - 1, m.R, m.R + 1, 2 ** 255, 2 ** 256 - 1, 2 ** 128, (1 << 16) - 1, 1 << 15, (1 << 15) + 1]


def _corrupt_point(rng, blob, off, width, pool_bad):
    kind = rng.randrange(4)
    b = bytearray(blob)
    if kind == 0:                                   # non-zero pad byte -> INVALID_ELEMENT
        b[off + 64 * rng.randrange(width // 64) + rng.randrange(16)] = rng.randrange(1, 256)
    elif kind == 1:                                 # coordinate >= p -> INVALID_ELEMENT
        c = rng.randrange(width // 64)
        b[off + 64 * c + 16:off + 64 * c + 64] = (m.P + rng.randrange(3)).to_bytes(48, "big")
    elif kind == 2:                                 # off-curve point -> NOT_ON_CURVE
        b[off:off + width] = m.encode_fp(rng.randrange(1, 1 << 64)) * (width // 64)
    else:                                           # on curve, outside the subgroup (legal for MSM)
        b[off:off + width] = rng.choice(pool_bad)
    return bytes(b)


def test_fuzz_g1_msm(X, clib):
    rng = random.Random(0x2537)
    prng = m.SplitMix64(99)
    ns = [m.encode_g1(m.random_g1(prng, False)) for _ in range(4)] + [m.encode_g1((0, 2))]
    for trial in range(40):
        n = rng.choice([1, 2, 3, 4, 5, 6, 17, 63, 64, 65, 100, 257, 300])
        base = clib.gen_msm_input("g1", n, rng.randrange(1, m.R), rng.randrange(1, m.R), rng.randrange(1 << 32))
        recs = [bytearray(base[i * 160:(i + 1) * 160]) for i in range(n)]
        for i in range(n):
            t = rng.random()
            if t < 0.08:
                recs[i][:128] = bytes(128)                                  # infinity
            elif t < 0.16 and i:
                recs[i][:128] = recs[rng.randrange(i)][:128]                # duplicate point
            elif t < 0.22 and i:
                src = recs[rng.randrange(i)]
                y = int.from_bytes(src[80:128], "big")
                recs[i][:128] = src[:64] + m.encode_fp((-y) % m.P if any(src[:128]) else 0)
            elif t < 0.27:
                recs[i][:128] = rng.choice(ns)
            if rng.random() < 0.2:
                recs[i][128:] = m.encode_scalar(rng.choice(SPECIAL_K))
            elif rng.random() < 0.1 and i:
                recs[i][128:] = recs[rng.randrange(i)][128:]
        inp = b"".join(bytes(r) for r in recs)
        if rng.random() < 0.35:
            inp = _corrupt_point(rng, inp, rng.randrange(n) * 160, 128, ns)
            if rng.random() < 0.5:
                inp = _corrupt_point(rng, inp, rng.randrange(n) * 160, 128, ns)
        assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp), trial


def test_fuzz_g2_msm(X, clib):
    rng = random.Random(0x2538)
    prng = m.SplitMix64(98)
    ns = [m.encode_g2(m.random_g2(prng, False)) for _ in range(3)]
    for trial in range(16):
        n = rng.choice([1, 2, 4, 5, 33, 64, 65, 130])
        base = clib.gen_msm_input("g2", n, rng.randrange(1, m.R), rng.randrange(1, m.R), rng.randrange(1 << 32))
        recs = [bytearray(base[i * 288:(i + 1) * 288]) for i in range(n)]
        for i in range(n):
            t = rng.random()
            if t < 0.1:
                recs[i][:256] = bytes(256)
            elif t < 0.2 and i:
                recs[i][:256] = recs[rng.randrange(i)][:256]
            elif t < 0.26:
                recs[i][:256] = rng.choice(ns)
            if rng.random() < 0.2:
                recs[i][256:] = m.encode_scalar(rng.choice(SPECIAL_K))
        inp = b"".join(bytes(r) for r in recs)
        if rng.random() < 0.35:
            inp = _corrupt_point(rng, inp, rng.randrange(n) * 288, 256, ns)
        assert call_x(X.g2_multiexp, inp) == clib.call("bls12_g2multiexp", inp), trial


def test_fuzz_pairing(X, clib):
    rng = random.Random(0x2539)
    prng = m.SplitMix64(97)
    ns1 = [m.encode_g1(m.random_g1(prng, False)) for _ in range(3)] + [m.encode_g1((0, 2))]
    ns2 = [m.encode_g2(m.random_g2(prng, False)) for _ in range(3)]
    G2e = m.encode_g2(m.G2)
    for trial in range(24):
        k = rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 64, 65, 100])
        a0, a1, b0, b1 = (rng.randrange(1, m.R) for _ in range(4))
        buf = bytearray(clib.gen_pairing_input(k, a0, a1, b0, b1))
        # optionally close the product to one with the last pair, optionally sprinkle infinities
        inf_at = set(i for i in range(k - 1) if rng.random() < 0.1)
        for i in inf_at:
            if rng.random() < 0.5:
                buf[i * 384:i * 384 + 128] = bytes(128)
            else:
                buf[i * 384 + 128:(i + 1) * 384] = bytes(256)
        if rng.random() < 0.6:
            s = sum(((a0 + i * a1) % m.R) * ((b0 + i * b1) % m.R) for i in range(k - 1) if i not in inf_at) % m.R
            delta = 0 if rng.random() < 0.7 else rng.randrange(1, 5)
            buf[(k - 1) * 384:] = m.encode_g1(m.g1_mul(m.G1, (delta - s) % m.R)) + G2e
        inp = bytes(buf)
        if rng.random() < 0.4:
            i = rng.randrange(k)
            if rng.random() < 0.5:
                inp = _corrupt_point(rng, inp, i * 384, 128, ns1)            # G1 side (non-subgroup -> code 2)
            else:
                inp = _corrupt_point(rng, inp, i * 384 + 128, 256, ns2)      # G2 side
            if rng.random() < 0.4:
                j = rng.randrange(k)                                         # a second bad pair: lowest index must win
                if rng.random() < 0.5:
                    inp = _corrupt_point(rng, inp, j * 384, 128, ns1)
                else:
                    inp = _corrupt_point(rng, inp, j * 384 + 128, 256, ns2)
        assert call_x(X.pairing, inp) == clib.call("bls12_pairing", inp), trial
