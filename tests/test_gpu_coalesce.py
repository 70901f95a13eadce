"""Coalescing of concurrent small multiexp calls: many threads call the reference ABI
at once with small inputs; the library serves the ones that queue up behind a busy engine with ONE
device pipeline over their concatenated records.  Every call must still get exactly its own bytes /
its own error code (a bad record in one call must not touch its neighbours), and batches must form."""
import threading

import pytest

import bls12_381 as m
from conftest import call_x

pytestmark = pytest.mark.gpu

A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321


def _hammer(jobs, rounds=3):
    """jobs: list of (fn, input, want).  One thread per job, all released together, several rounds."""
    bad = []
    gate = threading.Barrier(len(jobs))

    def work(fn, inp, want, idx):
        for _ in range(rounds):
            gate.wait()
            got = call_x(fn, inp)
            if got != want:
                bad.append((idx, got[0], want[0]))

    ths = [threading.Thread(target=work, args=(fn, inp, want, i)) for i, (fn, inp, want) in enumerate(jobs)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return bad


def test_concurrent_small_g1_calls_are_coalesced_and_exact(X, clib):
    sizes = [2, 3, 5, 8, 16, 17, 31, 64, 65, 100, 128, 129, 200, 256, 300, 512, 2, 7, 33, 90, 4, 6, 12, 48]
    jobs = []
    for i, n in enumerate(sizes):
        inp = clib.gen_msm_input("g1", n, A + i, B, 0x5150 + i)
        jobs.append((X.g1_multiexp, inp, clib.call("bls12_g1multiexp", inp)))
    # adversarial calls riding in the same batches
    P = m.g1_mul(m.G1, 0xabcdef)
    rec = lambda pt, k: m.encode_g1(pt) + m.encode_scalar(k)
    for inp in [rec(P, 7) + rec(m.ec_neg(m.FP, P), 7),                                   # sums to infinity
                b"".join(rec(None, 5) for _ in range(4)),                               # all infinity
                b"".join(rec(P, 2 ** 256 - 1) for _ in range(9)),                       # one bucket per window
                rec(P, 0) + rec((0, 2), 5) + rec(P, m.R)]:
        jobs.append((X.g1_multiexp, inp, clib.call("bls12_g1multiexp", inp)))
    # bad calls: their error code is theirs alone, lowest bad record first
    bad = bytearray(clib.gen_msm_input("g1", 40, A, B, 9))
    bad[30 * 160:30 * 160 + 128] = m.encode_fp(1) + m.encode_fp(1)                        # record 30 off curve
    jobs.append((X.g1_multiexp, bytes(bad), (1, None)))
    bad[7 * 160 + 2] = 1                                                                  # record 7 pad byte wins
    jobs.append((X.g1_multiexp, bytes(bad), (3, None)))
    # GPU route pinned: by default calls of up to 16 records run the library's host code (the crossover,
    # tests/test_gpu_routes.py) and would never reach the queue -- the tiny adversarial calls above are here
    # to ride in coalesced batches
    X.set_route(0)
    try:
        before = X.coalesce_stats()
        assert _hammer(jobs) == []
        pipelines, calls, largest = (b - a for a, b in zip(before, X.coalesce_stats()))
        queued = sum(1 for _, inp, _ in jobs if 2 <= len(inp) // 160 <= 512)    # above: straight to an engine
        assert calls == 3 * queued
        assert pipelines < calls, "no call ever shared a pipeline"
        assert X.coalesce_stats()[2] >= 2
    finally:
        X.set_route(-1)
    # default route: only the calls above the crossover are queued, the others are served on the host
    before = X.coalesce_stats()
    assert _hammer(jobs, rounds=1) == []
    calls = X.coalesce_stats()[1] - before[1]
    assert calls == sum(1 for _, inp, _ in jobs if 17 <= len(inp) // 160 <= 512)


def test_concurrent_small_g2_and_mixed_calls(X, clib):
    jobs = []
    for i, n in enumerate([3, 4, 9, 33, 64, 100, 5, 17]):
        inp = clib.gen_msm_input("g2", n, A + i, B, 0x6160 + i)
        jobs.append((X.g2_multiexp, inp, clib.call("bls12_g2multiexp", inp)))
    for i, n in enumerate([2, 50, 300, 1000, 3000]):                     # the last two bypass the queue
        inp = clib.gen_msm_input("g1", n, A + i, B, 0x7170 + i)
        jobs.append((X.g1_multiexp, inp, clib.call("bls12_g1multiexp", inp)))
    g2bad = bytearray(clib.gen_msm_input("g2", 6, A, B, 1))
    g2bad[3 * 288:3 * 288 + 256] = m.encode_fp(1) * 4
    jobs.append((X.g2_multiexp, bytes(g2bad), (1, None)))
    pr = clib.gen_pairing_input(4, 5, 7, 11, 13)
    jobs.append((X.pairing, pr, clib.call("bls12_pairing", pr)))
    assert _hammer(jobs) == []
    X.set_route(0)                                                       # everything on the GPU: the small G2 calls are queued too
    try:
        assert _hammer(jobs, rounds=2) == []
    finally:
        X.set_route(-1)


def _pairing_case(clib, k, seed, delta):
    """k pairs of generator multiples whose product is one iff delta == 0."""
    a0, a1, b0, b1 = 5 + seed, 7 + 2 * seed, 11 + 3 * seed, 13 + seed
    buf = bytearray(clib.gen_pairing_input(k, a0, a1, b0, b1))
    s = sum((a0 + i * a1) * (b0 + i * b1) for i in range(k - 1)) % m.R
    buf[(k - 1) * 384:] = m.encode_g1(m.g1_mul(m.G1, (delta - s) % m.R)) + m.encode_g2(m.G2)
    return bytes(buf)


def test_concurrent_small_pairing_calls_are_coalesced_and_exact(X, clib):
    rng = m.SplitMix64(21)
    jobs = []
    for i, k in enumerate([5, 6, 8, 9, 16, 17, 31, 32, 33, 48, 64, 5, 7, 12]):
        good = _pairing_case(clib, k, i, 0)
        jobs.append((X.pairing, good, (0, bytes(31) + b"\x01")))
        if i % 3 == 0:
            jobs.append((X.pairing, _pairing_case(clib, k, i, 1), (0, bytes(32))))
    inf = bytearray(_pairing_case(clib, 6, 3, 0))                 # pairs with a point at infinity contribute one
    jobs.append((X.pairing, bytes(inf[:5 * 384]) + m.encode_g1(None) + m.encode_g2(m.G2), clib.call("bls12_pairing", bytes(inf[:5 * 384]) + m.encode_g1(None) + m.encode_g2(m.G2))))
    # failing calls keep their own code: G1 subgroup at pair 4, G2 off curve at pair 2 (wins), pad byte in pair 0
    bad = bytearray(_pairing_case(clib, 8, 5, 0))
    bad[4 * 384:4 * 384 + 128] = m.encode_g1(m.random_g1(rng, False))
    jobs.append((X.pairing, bytes(bad), (2, None)))
    bad[2 * 384 + 128:2 * 384 + 384] = m.encode_fp(1) * 4
    jobs.append((X.pairing, bytes(bad), (1, None)))
    bad2 = bytearray(_pairing_case(clib, 20, 6, 0))
    bad2[3] = 9
    jobs.append((X.pairing, bytes(bad2), (3, None)))
    bad3 = bytearray(_pairing_case(clib, 10, 7, 0))
    bad3[9 * 384 + 128:9 * 384 + 384] = m.encode_g2(m.random_g2(rng, False))     # G2 subgroup, last pair
    jobs.append((X.pairing, bytes(bad3), (2, None)))
    for fn, inp, want in jobs[:6]:
        assert clib.call("bls12_pairing", inp) == want            # the oracle agrees with the constructed expectation
    before = X.coalesce_stats()
    assert _hammer(jobs, rounds=2) == []
    pipelines, calls, _ = (b - a for a, b in zip(before, X.coalesce_stats()))
    assert calls == 2 * len(jobs) and pipelines < calls
