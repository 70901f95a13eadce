"""One k-pair check through eip2537_hip_pairing_dev (input resident), 20 timed calls: ms per call and the stage intervals from the engine's HIP events."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blst_eip2537_amd as pkg
X = pkg.Eip2537Executor
X.init(0)
X.set_route(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = X.gen_pairing_input(k, 0x1234567890abcdef1234567890abcdef, 0xfedcba0987654321, 0x77, 0x99)
d = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
ts, dev, walk, aux0, aux1 = [], [], [], [], []
for it in range(25):
    t0 = time.perf_counter()
    out = X.dev_call("eip2537_hip_pairing_dev", d.data_ptr(), k)
    t1 = time.perf_counter()
    if it >= 5:
        ts.append((t1 - t0) * 1e3)
        tm = X.last_timing(); ax = X.last_timing_aux()
        dev.append(tm[0]); walk.append(tm[1]); aux0.append(ax[0]); aux1.append(ax[1])
med = statistics.median
print("k=%d  call ms min %.3f med %.3f | device %.3f walk %.3f membership %.3f walk-end->L %.3f" % (
    k, min(ts), med(ts), med(dev), med(walk), med(aux0), med(aux1)))
