"""bls12_g1multiexp on a 2^20-record HOST buffer (the reference ABI call): wall-clock per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blst_eip2537_amd as pkg
X = pkg.Eip2537Executor
if not os.environ.get("EIP2537_HIP_DEVICES"):
    X.init(0)
A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
grp = sys.argv[2] if len(sys.argv) > 2 else "g1"
n = 1 << log2n
inp = X.gen_msm_input(grp, n, A, B, (0x25370000 if grp == "g1" else 0x25370100) + log2n)
gp = os.path.join(ROOT, "tests", "golden", "%smsm_2p%d.hex" % (grp, log2n))
gold = bytes.fromhex(open(gp).read().strip()) if os.path.exists(gp) else None
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 23
ts = []
for i in range(reps):
    t0 = time.perf_counter()
    out = X.g1_multiexp(inp) if grp == "g1" else X.g2_multiexp(inp)
    ts.append((time.perf_counter() - t0) * 1e3)
w = ts[3:]
print(grp, "H2D_PIPELINE=%s H2D_STAGES=%s n=2^%d shards=%s golden_ok=%s host-ABI ms min %.3f med %.3f mean %.3f max %.3f (n=%d)" % (
      os.environ.get("EIP2537_H2D_PIPELINE", "default"), os.environ.get("EIP2537_H2D_STAGES", "default"), log2n, (X.last_plan() or {}).get("shards"),
      None if gold is None else out == gold, min(w), sorted(w)[len(w) // 2], sum(w) / len(w), max(w), len(w)), flush=True)
