"""64 concurrent callers of 128-record G2 multiexps (and G1) through the reference ABI: calls per second."""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blst_eip2537_amd import Eip2537Executor as X
for name, fn, inp in (("g2msm_128", "g2_multiexp", X.gen_msm_input("g2", 128, 3, 5, 7)), ("g1msm_128", "g1_multiexp", X.gen_msm_input("g1", 128, 3, 5, 7))):
    want = getattr(X, fn)(inp)
    for T in (16, 64):
        best = 0
        for rep in range(3):
            bad = []
            def run():
                for _ in range(40):
                    if getattr(X, fn)(inp) != want: bad.append(1)
            th = [threading.Thread(target=run) for _ in range(T)]
            t0 = time.perf_counter()
            [t.start() for t in th]; [t.join() for t in th]
            dt = time.perf_counter() - t0
            assert not bad
            best = max(best, T * 40 / dt)
        print("%s T=%d %.0f calls/s (G2_4LANE=%s)" % (name, T, best, os.environ.get("EIP2537_BATCH_G2_4LANE", "0")), flush=True)
