"""A/B of the two-level bucket reduce: G1 MSM 2^20 device-resident, golden check + timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blst_eip2537_amd as pkg
X = pkg.Eip2537Executor
X.init(0)
A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
inp = X.gen_msm_input("g1", n, A, B, 0x25370000 + log2n)
gp = os.path.join(ROOT, "tests", "golden", "g1msm_2p%d.hex" % log2n)
gold = bytes.fromhex(open(gp).read().strip()) if os.path.exists(gp) else None
d = torch.frombuffer(bytearray(inp), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
out = None
ts, ks = [], []
for i in range(12):
    t0 = time.perf_counter()
    out = X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(X.last_timing())
print("REDUCE_RC=%s n=2^%d golden_ok=%s  ms min %.3f med %.3f  pipeline %.3f accum %.3f" % (
    os.environ.get("EIP2537_REDUCE_RC", "1"), log2n, None if gold is None else out == gold, min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2],
    min(k[0] for k in ks[2:]), min(k[1] for k in ks[2:])), flush=True)
