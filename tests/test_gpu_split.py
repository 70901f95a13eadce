"""The multi-device split behind the reference ABI (reference src/eip2537.c:541-561 has one
CPU loop where this library cuts the record range over the listed devices, one host thread each).
A 1-GPU box rehearses it with EIP2537_HIP_DEVICES=0,0 -- two engine pools on the same GPU -- in a
fresh process (the device list is read once).  Checked bit-exact against the oracle / the analytic
goldens, including which error wins when bad records sit in different shards."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import oracle
from oracle import clib
import bls12_381 as m
from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error
mode = sys.argv[2]
A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321
def call(fn, inp):
    try: return 0, fn(inp)
    except Eip2537Error as e: return e.code, None
assert X.device_count() == int(sys.argv[3]), X.device_count()
if mode == "small":                      # EIP2537_HIP_SPLIT_MIN=64: every call below is cut into shards
    for n in (128, 129, 1000, 4097):
        g1 = clib.gen_msm_input("g1", n, A, B, 77 + n)
        assert call(X.g1_multiexp, g1) == clib.call("bls12_g1multiexp", g1), ("g1", n)
    for n in (130, 777):
        g2 = clib.gen_msm_input("g2", n, A, B, 99 + n)
        assert call(X.g2_multiexp, g2) == clib.call("bls12_g2multiexp", g2), ("g2", n)
    # partial sums that cancel across shards / a shard that is all infinity
    P = m.g1_mul(m.G1, 0xabcdef)
    rec = lambda pt, k: m.encode_g1(pt) + m.encode_scalar(k)
    inp = b"".join(rec(P, 5) for _ in range(64)) + b"".join(rec(m.ec_neg(m.FP, P), 5) for _ in range(64))
    assert call(X.g1_multiexp, inp) == (0, bytes(128))
    inp = b"".join(rec(None, 9) for _ in range(64)) + b"".join(rec(P, 1) for _ in range(64))
    assert call(X.g1_multiexp, inp) == clib.call("bls12_g1multiexp", inp)
    # error order over shards: the lowest-index bad record wins, whatever shard it is in
    good = bytearray(clib.gen_msm_input("g1", 256, A, B, 5))
    bad = bytearray(good); bad[200 * 160 + 0] = 1                    # pad byte: INVALID_ELEMENT in shard 1
    assert call(X.g1_multiexp, bytes(bad)) == (3, None)
    bad[10 * 160 + 16:10 * 160 + 128] = m.encode_g1((1, 1))[16:]     # (1,1) off curve in shard 0 -> wins
    assert call(X.g1_multiexp, bytes(bad)) == (1, None)
    # pairing: product over shards, and the error order
    k = 160
    a0, a1, b0, b1 = 5, 7, 11, 13
    pairs = bytearray(X.gen_pairing_input(k, a0, a1, b0, b1))
    s = sum((a0 + i * a1) * (b0 + i * b1) for i in range(k - 1)) % m.R
    pairs[(k - 1) * 384:] = m.encode_g1(m.g1_mul(m.G1, (-s) % m.R)) + m.encode_g2(m.G2)
    assert call(X.pairing, bytes(pairs)) == (0, bytes(31) + b"\x01")
    tw = bytearray(pairs); tw[(k - 1) * 384:(k - 1) * 384 + 128] = m.encode_g1(m.g1_mul(m.G1, (1 - s) % m.R))
    assert call(X.pairing, bytes(tw)) == (0, bytes(32))
    rng = m.SplitMix64(3)
    e = bytearray(pairs)
    e[150 * 384:150 * 384 + 128] = m.encode_g1(m.random_g1(rng, False))          # non-subgroup G1, shard 1
    assert call(X.pairing, bytes(e)) == (2, None)
    e[20 * 384 + 128:20 * 384 + 384] = m.encode_g2((((1, 0)), ((1, 0))))         # off-curve G2, shard 0 -> wins
    assert call(X.pairing, bytes(e)) == (1, None)
    assert X.last_plan() is not None
elif mode == "pipe":                     # ONE device.  G1 (round 4): EIP2537_H2D_STAGES=1,2,2 stages every c = 16 call as three record shards
    n = (1 << 17) + 1                    # into ONE bucket space (msm.hip: StagedCopy, bucket accumulators); G2: EIP2537_H2D_PIPELINE=3, three
    g1 = clib.gen_msm_input("g1", n, A, B, 4242)       # independent shard pipelines whose copies follow each other (CopyGate)
    assert call(X.g1_multiexp, g1) == clib.call("bls12_g1multiexp", g1)
    assert X.last_plan()["units"] == n and X.last_plan()["shards"] == 3, X.last_plan()
    bad = bytearray(g1); bad[120000 * 160 + 0] = 1                     # pad byte: INVALID_ELEMENT in shard 2
    assert call(X.g1_multiexp, bytes(bad)) == (3, None)
    bad[10 * 160 + 16:10 * 160 + 128] = m.encode_g1((1, 1))[16:]       # (1,1) off curve in shard 0 -> wins
    assert call(X.g1_multiexp, bytes(bad)) == (1, None)
    # buckets shared between shards: the same point with the same scalar in every shard (doubling inside a bucket accumulator),
    # a shard that cancels an earlier one (accumulators fall back to infinity), a shard of infinities
    P = m.g1_mul(m.G1, 0xabcdef)
    rec = lambda pt, k: m.encode_g1(pt) + m.encode_scalar(k)
    third = n // 5 + 7
    body = bytearray(g1)
    for i in range(3): body[(i * 52429 + 5) * 160:(i * 52429 + 6) * 160] = rec(P, 0x1234567)
    assert call(X.g1_multiexp, bytes(body)) == clib.call("bls12_g1multiexp", bytes(body))
    canc = bytearray(g1[:third * 160]) + bytearray(len(g1) - third * 160)
    for i in range(third, n):                                           # records third.. : infinity, except a mirror of shard 0's first records
        j = i - third
        if j < third: canc[i * 160:i * 160 + 160] = g1[j * 160:j * 160 + 64] + m.encode_g1(m.ec_neg(m.FP, m.decode_g1(g1[j * 160:j * 160 + 128])))[64:] + g1[j * 160 + 128:j * 160 + 160]
    assert call(X.g1_multiexp, bytes(canc)) == (0, bytes(128))
    gold = lambda name: bytes.fromhex(open(os.path.join(sys.argv[1], "tests", "golden", name)).read().strip())
    g2 = X.gen_msm_input("g2", 1 << 16, A, B, 0x25370100 + 16)
    assert call(X.g2_multiexp, g2) == (0, gold("g2msm_2p16.hex"))
    assert X.last_plan()["units"] in ((1 << 16) // 3, (1 << 16) // 3 + 1), X.last_plan()
    # concurrent staged calls: each holds one engine slot and its helper thread
    import threading
    want = clib.call("bls12_g1multiexp", g1)
    res = [None] * 6
    def worker(i): res[i] = call(X.g1_multiexp, g1)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(r == want for r in res)
elif mode == "dev_shards":               # ONE device, DEVICE-resident input, EIP2537_DEV_STAGES=1,2,2: three record shards of one bucket space,
    import torch                         # the sort stage of shard s + 1 on a second stream beside the accumulate of shard s (msm.hip)
    dev = lambda buf: torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    def dcall(buf, n):
        d = dev(buf)
        try: return 0, X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
        except Eip2537Error as e: return e.code, None
    n = (1 << 17) + 1
    g1 = clib.gen_msm_input("g1", n, A, B, 4242)
    assert dcall(g1, n) == clib.call("bls12_g1multiexp", g1)
    assert X.last_plan()["units"] == n and X.last_plan()["shards"] == 3, X.last_plan()
    for rep in range(3): assert dcall(g1, n) == clib.call("bls12_g1multiexp", g1)         # the doubled buffers and events are reused
    bad = bytearray(g1); bad[120000 * 160 + 0] = 1                     # pad byte: INVALID_ELEMENT in shard 2
    assert dcall(bytes(bad), n) == (3, None)
    bad[10 * 160 + 16:10 * 160 + 128] = m.encode_g1((1, 1))[16:]       # (1,1) off curve in shard 0 -> wins
    assert dcall(bytes(bad), n) == (1, None)
    P = m.g1_mul(m.G1, 0xabcdef)
    rec = lambda pt, k: m.encode_g1(pt) + m.encode_scalar(k)
    body = bytearray(g1)
    for i in range(3): body[(i * 52429 + 5) * 160:(i * 52429 + 6) * 160] = rec(P, 0x1234567)     # the same bucket in every shard
    assert dcall(bytes(body), n) == clib.call("bls12_g1multiexp", bytes(body))
    third = n // 5 + 7
    canc = bytearray(g1[:third * 160]) + bytearray(len(g1) - third * 160)
    for i in range(third, n):                                           # a later shard cancels shard 0; the rest is infinity
        j = i - third
        if j < third: canc[i * 160:i * 160 + 160] = g1[j * 160:j * 160 + 64] + m.encode_g1(m.ec_neg(m.FP, m.decode_g1(g1[j * 160:j * 160 + 128])))[64:] + g1[j * 160 + 128:j * 160 + 160]
    assert dcall(bytes(canc), n) == (0, bytes(128))
    same = b"".join(rec(P, 7) for _ in range(n))                       # degenerate input: the sort stands down, the host re-runs unsharded
    assert dcall(same, n) == clib.call("bls12_g1multiexp", same)
    gold = lambda name: bytes.fromhex(open(os.path.join(sys.argv[1], "tests", "golden", name)).read().strip())
    g20 = X.gen_msm_input("g1", 1 << 20, A, B, 0x25370000 + 20)
    assert dcall(g20, 1 << 20) == (0, gold("g1msm_2p20.hex"))
    assert X.last_plan()["shards"] == 3
elif mode == "pipe_default":             # ONE device, default policy: 2^20 G1 records staged in several shards, 2^16 = one copy
    gold = lambda name: bytes.fromhex(open(os.path.join(sys.argv[1], "tests", "golden", name)).read().strip())
    g1 = X.gen_msm_input("g1", 1 << 20, A, B, 0x25370000 + 20)
    assert call(X.g1_multiexp, g1) == (0, gold("g1msm_2p20.hex"))
    assert X.last_plan()["units"] == 1 << 20 and X.last_plan()["shards"] >= 3, X.last_plan()
    g1s = X.gen_msm_input("g1", 1 << 16, A, B, 0x25370000 + 16)
    assert call(X.g1_multiexp, g1s) == (0, gold("g1msm_2p16.hex"))
    assert X.last_plan()["units"] == 1 << 16 and X.last_plan()["shards"] == 1
else:                                    # default thresholds at the BASELINE sizes
    gold = lambda name: bytes.fromhex(open(os.path.join(sys.argv[1], "tests", "golden", name)).read().strip())
    g1 = X.gen_msm_input("g1", 1 << 20, A, B, 0x25370000 + 20)
    assert call(X.g1_multiexp, g1) == (0, gold("g1msm_2p20.hex"))
    p, _ = X.last_timing()
    assert X.last_plan()["units"] == 1 << 19 and X.last_plan()["shards"] >= 2, X.last_plan()     # each device got half the records, staged in shards
    g2 = X.gen_msm_input("g2", 1 << 16, A, B, 0x25370100 + 16)
    assert call(X.g2_multiexp, g2) == (0, gold("g2msm_2p16.hex"))
    assert X.last_plan()["units"] == 1 << 15, X.last_plan()
    # below two shards' worth nothing is cut
    g1s = X.gen_msm_input("g1", 1 << 16, A, B, 0x25370000 + 16)
    assert call(X.g1_multiexp, g1s) == (0, gold("g1msm_2p16.hex"))
    assert X.last_plan()["units"] == 1 << 16
print("split worker ok")
'''


def _run(tmp_path, mode, devices, extra_env):
    script = tmp_path / "split_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, EIP2537_HIP_DEVICES=devices, **extra_env)
    env.pop("EIP2537_HIP_DEVICE", None)
    cp = subprocess.run([sys.executable, str(script), ROOT, mode, str(len(devices.split(",")))], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert cp.returncode == 0 and "split worker ok" in cp.stdout, cp.stdout[-3000:]


def test_split_small_inputs_two_pools_one_gpu(tmp_path, clib, X):
    _run(tmp_path, "small", "0,0", {"EIP2537_HIP_SPLIT_MIN": "64"})


def test_split_three_pools_ragged(tmp_path, clib, X):
    _run(tmp_path, "small", "0,0,0", {"EIP2537_HIP_SPLIT_MIN": "40"})


def test_split_baseline_sizes_default_thresholds(tmp_path, clib, X):
    _run(tmp_path, "full", "0,0", {})


def test_pipelined_host_input_three_shards_one_device(tmp_path, clib, X):
    _run(tmp_path, "pipe", "0", {"EIP2537_H2D_PIPELINE": "3", "EIP2537_H2D_STAGES": "1,2,2"})


def test_pipelined_host_input_default_policy(tmp_path, clib, X):
    _run(tmp_path, "pipe_default", "0", {})


def test_device_ordinal_out_of_range_is_an_error(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error\n"
            "try:\n    X.init(63)\n    print('accepted')\nexcept Eip2537Error as e:\n    print('refused', e.code)\n" % ROOT)
    cp = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "refused 7" in cp.stdout, cp.stdout


def test_callers_current_device_is_preserved(X, clib):
    import torch
    before = torch.cuda.current_device()
    inp = clib.gen_msm_input("g1", 100, 3, 5, 1)
    X.g1_multiexp(inp)
    assert torch.cuda.current_device() == before
    assert X.trim(0) >= 0                                  # idle slots give their workspace back
    assert X.g1_multiexp(inp) == clib.call("bls12_g1multiexp", inp)[1]


def test_device_resident_record_shards(tmp_path, clib, X):
    _run(tmp_path, "dev_shards", "0", {"EIP2537_DEV_STAGES": "1,2,2"})
