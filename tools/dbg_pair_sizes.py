"""Debug helper: bls12_pairing on the GPU route for a range of batch sizes whose product is 1 by construction
(and 'not 1' when the last pair is shifted), so that every branch of the product tree is exercised."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa
import bls12_381 as m
import blst_eip2537_amd as pkg
X = pkg.Eip2537Executor
X.init(0)
X.set_route(0)
A, B = 0x1234567890abcdef1234567890abcdef, 0xfedcba0987654321

def batch(k, delta):
    a0, a1, b0, b1 = A, B, B ^ 0x55, A ^ 0x33
    buf = X.gen_pairing_input(k, a0, a1, b0, b1)
    s = sum(((a0 + i * a1) % m.R) * ((b0 + i * b1) % m.R) for i in range(k - 1)) % m.R
    last = m.encode_g1(m.g1_mul(m.G1, (delta - s) % m.R)) + m.encode_g2(m.G2)
    return buf[:-384] + last

sizes = [int(a) for a in sys.argv[1:]] or [2, 5, 63, 64, 65, 66, 127, 128, 129, 200, 448, 449, 450, 600, 896, 897, 1024, 2000, 4096, 5000]
for k in sizes:
    good, bad = batch(k, 0), batch(k, 1)
    g, b = X.pairing(good), X.pairing(bad)
    print(k, "good ->", g[-1], "bad ->", b[-1], "OK" if (g[-1] == 1 and b[-1] == 0) else "WRONG", flush=True)
