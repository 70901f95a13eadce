"""The launch shapes are derived from the device's compute-unit count (csrc/msm.hip: chip_shape), not from the 256 CUs of a whole MI355X
(VERDICT r3 item 8).  A fresh process with EIP2537_HIP_CUS=128 -- what a partition or a CU mask would report -- must still be bit-exact on
every plan whose shape depends on the count: the two-level reduce (32-bucket row / column chains instead of 16), the 4-lane / 8-lane reduce
grids, the pairing walk's whole-SIMD threshold and the fold grid."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import oracle
from oracle import clib
from blst_eip2537_amd import Eip2537Executor as X, Eip2537Error
A, B = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6, 0x0123456789abcdef0fedcba987654321
gold = lambda name: bytes.fromhex(open(os.path.join(sys.argv[1], "tests", "golden", name)).read().strip())
n = (1 << 17) + 3                                    # c = 16 plan: two-level reduce
g1 = clib.gen_msm_input("g1", n, A, B, 31337)
assert (0, X.g1_multiexp(g1)) == clib.call("bls12_g1multiexp", g1)
assert X.last_plan()["window_bits"] == 16, X.last_plan()
g1b = X.gen_msm_input("g1", 1 << 16, A, B, 0x25370000 + 16)      # c = 13: 4-lane reduce grid
assert X.g1_multiexp(g1b) == gold("g1msm_2p16.hex")
g2 = X.gen_msm_input("g2", 1 << 16, A, B, 0x25370100 + 16)       # 8-lane reduce grid
assert X.g2_multiexp(g2) == gold("g2msm_2p16.hex")
import bls12_381 as m
for k in (700, 4096):                                # below / above the halved whole-SIMD threshold
    a0, a1, b0, b1 = 5, 7, 11, 13
    pairs = bytearray(X.gen_pairing_input(k, a0, a1, b0, b1))
    s = sum((a0 + i * a1) * (b0 + i * b1) for i in range(k - 1)) % m.R
    pairs[(k - 1) * 384:] = m.encode_g1(m.g1_mul(m.G1, (-s) % m.R)) + m.encode_g2(m.G2)
    assert X.pairing(bytes(pairs)) == bytes(31) + b"\x01", k
    pairs[(k - 1) * 384:(k - 1) * 384 + 128] = m.encode_g1(m.g1_mul(m.G1, (1 - s) % m.R))
    assert X.pairing(bytes(pairs)) == bytes(32), k
big = X.gen_msm_input("g1", 1 << 20, A, B, 0x25370000 + 20)
assert X.g1_multiexp(big) == gold("g1msm_2p20.hex")
t0 = time.perf_counter()
for _ in range(3): X.g1_multiexp(big)
ms = (time.perf_counter() - t0) / 3 * 1e3
assert ms < 40.0, ms                                 # a sane time: the whole chip is there, only the launch shapes assume half of it
print("chip shape worker ok: 2^20 host call %.2f ms with EIP2537_HIP_CUS=%s" % (ms, os.environ.get("EIP2537_HIP_CUS")))
'''


def test_half_chip_override(tmp_path, clib, X):
    script = tmp_path / "chip_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, EIP2537_HIP_CUS="128")
    cp = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert cp.returncode == 0 and "chip shape worker ok" in cp.stdout, cp.stdout[-3000:]
    assert "128 compute units" in cp.stdout, cp.stdout[-3000:]          # the library says so once
