"""Experiment: ONE 2^20-record G1 MSM on resident input as K concurrent shard pipelines (eip2537_hip_g1msm_partial_dev from K
host threads, then the combine) against the single pipeline (eip2537_hip_g1multiexp_dev)."""
import os, sys, time, threading, statistics
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
gold = bytes.fromhex(open(os.path.join(ROOT, "tests", "golden", "g1msm_2p%d.hex" % log2n)).read().strip())
d = torch.frombuffer(bytearray(inp), dtype=torch.uint8).cuda()
torch.cuda.synchronize()

def whole():
    return X.dev_call("eip2537_hip_g1multiexp_dev", d.data_ptr(), n)

def sharded(k, stagger_us=0):
    parts = [None] * k
    def run(s):
        if stagger_us: time.sleep(s * stagger_us * 1e-6)
        lo, hi = n * s // k, n * (s + 1) // k
        parts[s] = X.dev_call("eip2537_hip_g1msm_partial_dev", d.data_ptr() + lo * 160, hi - lo)
    th = [threading.Thread(target=run, args=(s,)) for s in range(1, k)]
    [t.start() for t in th]
    run(0)
    [t.join() for t in th]
    return X.combine("eip2537_hip_g1msm_combine", parts)

def timeit(fn, reps=15):
    for _ in range(3): out = fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return out, min(ts), statistics.median(ts)

out, mn, md = timeit(whole)
print("whole           ok=%s  min %.3f med %.3f ms" % (out == gold, mn, md), flush=True)
for k in (2, 3, 4):
    for st in (0, 300):
        out, mn, md = timeit(lambda: sharded(k, st))
        print("shards k=%d stagger %3d us  ok=%s  min %.3f med %.3f ms" % (k, st, out == gold, mn, md), flush=True)
out, mn, md = timeit(whole)
print("whole           ok=%s  min %.3f med %.3f ms" % (out == gold, mn, md), flush=True)
