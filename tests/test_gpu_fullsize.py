"""GPU parity at the BASELINE.json sizes (configs 2-5): bit-exact against the CPU oracle where the
oracle finishes in seconds, against the committed analytic goldens (tests/golden/, produced by the
Python model), and through size-independent properties: shard -> partial -> combine equals the
whole, record order does not matter, device-pointer and host-pointer entry points agree."""
import os
import random

import pytest
import torch

import bls12_381 as m
from conftest import ROOT, call_x

pytestmark = pytest.mark.gpu

A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321


def _gold(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return bytes.fromhex(f.read().strip())


def _dev(buf):
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


def test_config2_g1_msm_2p16(X, clib):
    inp = X.gen_msm_input("g1", 1 << 16, A, B, 0x25370000 + 16)
    got = call_x(X.g1_multiexp, inp)
    assert got == (0, _gold("g1msm_2p16.hex"))
    assert got == clib.call("bls12_g1multiexp", inp)          # oracle: Bos-Coster, ~2 s
    d = _dev(inp)
    assert X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), 1 << 16) == got[1]


def test_config3_g2_msm_2p16(X, clib):
    inp = X.gen_msm_input("g2", 1 << 16, A, B, 0x25370100 + 16)
    got = call_x(X.g2_multiexp, inp)
    assert got == (0, _gold("g2msm_2p16.hex"))
    assert got == clib.call("bls12_g2multiexp", inp)


def _pairing_batch(X, k, delta):
    a0, a1, b0, b1 = A, B, B ^ 0x55, A ^ 0x33
    buf = X.gen_pairing_input(k, a0, a1, b0, b1)
    s = sum(((a0 + i * a1) % m.R) * ((b0 + i * b1) % m.R) for i in range(k - 1)) % m.R
    last = m.encode_g1(m.g1_mul(m.G1, (delta - s) % m.R)) + m.encode_g2(m.G2)
    return buf[:-384] + last


def test_config4_pairing_2p12(X, clib):
    k = 1 << 12
    good, bad = _pairing_batch(X, k, 0), _pairing_batch(X, k, 1)
    assert call_x(X.pairing, good) == (0, bytes(31) + b"\x01")
    assert call_x(X.pairing, bad) == (0, bytes(32))
    assert clib.call("bls12_pairing", good) == (0, bytes(31) + b"\x01")      # oracle, ~3 s
    # error order at scale: non-subgroup G1 at pair 3000, off-curve G2 at pair 1234 -> pair 1234 wins
    rng = m.SplitMix64(3)
    t = bytearray(good)
    t[3000 * 384:3000 * 384 + 128] = m.encode_g1(m.random_g1(rng, False))
    assert call_x(X.pairing, bytes(t)) == (2, None)
    t[1234 * 384 + 128:1234 * 384 + 384] = m.encode_fp(1) * 4
    assert call_x(X.pairing, bytes(t)) == (1, None)
    # sharded: 4 partial Miller products combined == whole
    d = _dev(good)
    parts = [X.dev_call("eip2537_hip_pairing_partial_dev", d.data_ptr() + s * 1024 * 384, 1024) for s in range(4)]
    assert X.combine("eip2537_hip_pairing_combine", parts) == bytes(31) + b"\x01"


def test_config5_g1_msm_2p20_whole_and_sharded(X):
    n = 1 << 20
    inp = X.gen_msm_input("g1", n, A, B, 0x25370000 + 20)
    gold = _gold("g1msm_2p20.hex")
    d = _dev(inp)
    assert X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n) == gold
    # BASELINE config 5 shape on one GPU: 8 contiguous shards -> 8 partials -> combine
    per = n // 8
    parts = [X.dev_call("eip2537_hip_g1msm_partial_dev", d.data_ptr() + s * per * 160, per) for s in range(8)]
    assert X.combine("eip2537_hip_g1msm_combine", parts) == gold
    # ragged shards give the same answer
    cuts = [0, 1, 4097, 300000, 300001, 777777, n]
    parts = [X.dev_call("eip2537_hip_g1msm_partial_dev", d.data_ptr() + lo * 160, hi - lo) for lo, hi in zip(cuts, cuts[1:])]
    assert X.combine("eip2537_hip_g1msm_combine", parts) == gold


def test_msm_is_order_independent_and_linear(X, clib):
    n = 5000
    inp = X.gen_msm_input("g1", n, A, B, 77)
    recs = [inp[i * 160:(i + 1) * 160] for i in range(n)]
    whole = X.g1_multiexp(inp)
    random.Random(1).shuffle(recs)
    assert X.g1_multiexp(b"".join(recs)) == whole
    # MSM(first half) + MSM(second half) == MSM(all), via the add precompile
    h1, h2 = X.g1_multiexp(inp[:2500 * 160]), X.g1_multiexp(inp[2500 * 160:])
    assert X.g1_add(h1 + h2) == whole
    # scaling every scalar by 2 (mod 2^256 wrap avoided: use small scalars) doubles the result
    small = b"".join(r[:128] + m.encode_scalar(i + 1) for i, r in enumerate(recs[:300]))
    twice = b"".join(r[:128] + m.encode_scalar(2 * (i + 1)) for i, r in enumerate(recs[:300]))
    s1 = X.g1_multiexp(small)
    assert X.g1_multiexp(twice) == X.g1_add(s1 + s1)
    assert clib.call("bls12_g1multiexp", small) == (0, s1)


def test_g2_msm_sharded_equals_whole(X, clib):
    n = 3000
    inp = X.gen_msm_input("g2", n, A, B, 78)
    whole = X.g2_multiexp(inp)
    assert clib.call("bls12_g2multiexp", inp) == (0, whole)
    d = _dev(inp)
    parts = [X.dev_call("eip2537_hip_g2msm_partial_dev", d.data_ptr() + s * 1000 * 288, 1000) for s in range(3)]
    assert X.combine("eip2537_hip_g2msm_combine", parts) == whole


def test_heavy_buckets_top_window_and_equal_scalars(X, clib):
    """Sizes whose window plan leaves a 1-3 bit top window (its buckets hold ~n/2 records) and the
    adversarial all-equal-scalars input (one bucket per window holds everything): multi-task buckets
    are folded in parallel (k_msm_fold) and must still give the oracle's bytes."""
    inp = X.gen_msm_input("g1", 1 << 14, A, B, 1414)
    assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp)
    # 2^18 (c = 15, 1-bit top window): whole == 4 shards of 2^16 (c = 13) == ragged shards
    n = 1 << 18
    inp = X.gen_msm_input("g1", n, A, B, 1818)
    d = _dev(inp)
    whole = X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    parts = [X.dev_call("eip2537_hip_g1msm_partial_dev", d.data_ptr() + s * (n // 4) * 160, n // 4) for s in range(4)]
    assert X.combine("eip2537_hip_g1msm_combine", parts) == whole
    cuts = [0, 5000, 70000, n]
    parts = [X.dev_call("eip2537_hip_g1msm_partial_dev", d.data_ptr() + lo * 160, hi - lo) for lo, hi in zip(cuts, cuts[1:])]
    assert X.combine("eip2537_hip_g1msm_combine", parts) == whole
    # all scalars equal: sum k*P_i = k * sum P_i ; 20000 records, k = 2^256 - 1 and k = 1
    m20 = 20000
    base = X.gen_msm_input("g1", m20, A, B, 7)
    for k in (2 ** 256 - 1, 1, m.R + 5):
        same = b"".join(base[i * 160:i * 160 + 128] + m.encode_scalar(k) for i in range(m20))
        got = X.g1_multiexp(same)
        if k == 1:
            ones = got
        assert clib.call("bls12_g1multiexp", same) == (0, got), k
    assert X.g1_mul(ones + m.encode_scalar(2 ** 256 - 1)) == X.g1_multiexp(
        b"".join(base[i * 160:i * 160 + 128] + m.encode_scalar(2 ** 256 - 1) for i in range(m20)))
    # the same shapes through the plan of the large sizes (c = 16: limb-form accumulate, fold and reduce,
    # csrc/limb30.h): one bucket per window with all 20000 records (k_msm_fold_big), and scalars from a set
    # of 97 values, ~200 records = 4 tasks per bucket (k_msm_fold_small)
    X.set_window(16)
    try:
        same = b"".join(base[i * 160:i * 160 + 128] + m.encode_scalar(2 ** 256 - 1) for i in range(m20))
        assert clib.call("bls12_g1multiexp", same) == (0, X.g1_multiexp(same))
        assert X.last_plan()["kernel"] == ("k_msm_accum<eip::Fp>" if os.environ.get("EIP2537_LIMB_FORM") == "0" else "k_msm_accum_l")
        rng = m.SplitMix64(97)
        pool = [rng.scalar256() for _ in range(97)]
        few = b"".join(base[i * 160:i * 160 + 128] + m.encode_scalar(pool[(i * 31) % 97]) for i in range(m20))
        assert clib.call("bls12_g1multiexp", few) == (0, X.g1_multiexp(few))
        # the partitioned sort of the c = 16 plans (k_sort_*): a ragged second slice (32 768 + 7 233 records), and G2
        rag = X.gen_msm_input("g1", 40001, A, B, 4001)
        assert clib.call("bls12_g1multiexp", rag) == (0, X.g1_multiexp(rag))
        # degenerate digit patterns through the same plan: no entries at all, one entry, one bucket per window, two buckets
        recs = [rag[i * 160:(i + 1) * 160] for i in range(40001)]
        for name, inp in {
            "all scalars zero": b"".join(r[:128] + bytes(32) for r in recs),
            "all points infinity": b"".join(bytes(128) + r[128:] for r in recs),
            "one live record": b"".join((r if i == 12345 else r[:128] + bytes(32)) for i, r in enumerate(recs)),
            "scalars = 2^255": b"".join(r[:128] + m.encode_scalar(1 << 255) for r in recs),
            "two values alternating": b"".join(r[:128] + m.encode_scalar((1 << 200) + 5 if i & 1 else 2 ** 256 - 3) for i, r in enumerate(recs)),
        }.items():
            assert call_x(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp), name
        g2 = X.gen_msm_input("g2", 3001, A, B, 3001)
        assert clib.call("bls12_g2multiexp", g2) == (0, X.g2_multiexp(g2))
        assert X.last_plan()["window_bits"] == 16
    finally:
        X.set_window(0)
    # G2 with a heavy top window
    inp = X.gen_msm_input("g2", 1 << 12, A, B, 1212)
    assert call_x(X.g2_multiexp, inp) == clib.call("bls12_g2multiexp", inp)


def test_pairing_all_walk_kernels_ragged_sizes(X, clib):
    """Ragged sizes around every switch of the pairing pipeline: up to 64 pairs one line-product block per step (a quad per
    line), up to 448 pairs one line per quad over 2..7 blocks, beyond that quads fold several lines; walk + membership
    waves claim whole SIMDs while they fit the chip (k / 8 + k / 16 <= 1024, i.e. up to 5461 pairs).  Each size closed to
    the identity and broken by one, and an error in the last (partial) group."""
    for k in (63, 65, 447, 449, 2100, 5461, 5463):
        good, bad = _pairing_batch(X, k, 0), _pairing_batch(X, k, 1)
        assert call_x(X.pairing, good) == (0, bytes(31) + b"\x01"), k
        assert call_x(X.pairing, bad) == (0, bytes(32)), k
        t = bytearray(good)
        t[(k - 2) * 384 + 128:(k - 2) * 384 + 384] = m.encode_fp(1) * 4       # G2 off curve in pair k-2
        assert call_x(X.pairing, bytes(t)) == (1, None), k
    assert clib.call("bls12_pairing", _pairing_batch(X, 2100, 0)) == (0, bytes(31) + b"\x01")
