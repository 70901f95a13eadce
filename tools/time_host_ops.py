import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "pymodel"))
from blst_eip2537_amd import Eip2537Executor as X
import bls12_381 as m
P2 = m.g2_mul(m.G2, 0xabcdef12345)
P1 = m.g1_mul(m.G1, 0xabcdef12345)
k = m.encode_scalar(0xf3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e61234567)
g2in = m.encode_g2(P2) + k
g1in = m.encode_g1(P1) + k
fp2in = bytes(16) + (12345).to_bytes(48, "big") + bytes(16) + (6789).to_bytes(48, "big")
def t(fn, arg, n=200):
    fn(arg)
    t0 = time.perf_counter()
    for _ in range(n): fn(arg)
    return (time.perf_counter() - t0) / n * 1e6
print("HOST_IFMA=%s  g1mul %.1f us  g2mul %.1f us  map_fp2_to_g2 %.1f us" % (os.environ.get("EIP2537_HOST_IFMA", "1"), t(X.g1_mul, g1in), t(X.g2_mul, g2in), t(X.map_fp2_to_g2, fp2in)))
