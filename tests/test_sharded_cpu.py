"""N > 1 path on CPU: two gloo ranks shard one MSM / one pairing batch by record range, exchange
their partials with all_gather and combine them with the product's host-side combine functions
(include/eip2537_hip.h).  The per-shard partials come from the CPU oracle here (no GPU in this
container); on the GPU box the same exchange runs in bench.py over RCCL."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import oracle
from oracle import clib
import bls12_381 as m
from blst_eip2537_amd import Eip2537Executor as X

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
RM = (1 << 384) % m.P
mont = lambda v: (v * RM % m.P).to_bytes(48, "little")
def g1_partial(enc):            # affine (x,y) -> XYZZ partial (x, y, 1, 1); infinity -> zz = 0
    if enc == bytes(128): return bytes(192)
    x, y = int.from_bytes(enc[16:64], "big"), int.from_bytes(enc[80:128], "big")
    return mont(x) + mont(y) + mont(1) + mont(1)
A, B, SEED, N = 0x1f3a5c7e9b2d, 0x123456789abcdef, 4242, 600
per = N // world
full = X.gen_msm_input("g1", N, A, B, SEED)
mine = X.gen_msm_input("g1", per, A, B, SEED, start=rank * per)     # each rank builds only its shard
assert mine == full[rank * per * 160:(rank + 1) * per * 160]
rc, part_aff = clib.call("bls12_g1multiexp", mine)
assert rc == 0
t = torch.frombuffer(bytearray(g1_partial(part_aff)), dtype=torch.uint8)
gathered = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
out = X.combine("eip2537_hip_g1msm_combine", [bytes(g.numpy().tobytes()) for g in gathered])
rc, want = clib.call("bls12_g1multiexp", full)
assert rc == 0 and out == want, "sharded MSM != whole MSM"
# pairing: shard the pairs, exchange Fp12 Miller products, one final exponentiation
K = 8
a0, a1, b0, b1 = 5, 7, 11, 13
pairs = bytearray(X.gen_pairing_input(K, a0, a1, b0, b1))
s = sum((a0 + i * a1) * (b0 + i * b1) for i in range(K - 1)) % m.R
pairs[(K - 1) * 384:] = m.encode_g1(m.g1_mul(m.G1, (-s) % m.R)) + m.encode_g2(m.G2)
kp = K // world
shard = bytes(pairs[rank * kp * 384:(rank + 1) * kp * 384])
acc = m.F12_ONE
for o in range(0, len(shard), 384):          # Miller product of the shard, via the oracle, as Montgomery words
    raw = clib.pairing_fp12(shard[o:o + 384], final_exp=False)
    coeffs = [int.from_bytes(raw[i * 48:(i + 1) * 48], "big") for i in range(12)]
    f = tuple(tuple((coeffs[c6 * 6 + c2 * 2], coeffs[c6 * 6 + c2 * 2 + 1]) for c2 in range(3)) for c6 in range(2))
    acc = m.f12_mul(acc, f)
blob = b"".join(mont(c) for c6 in acc for c2 in c6 for c in c2)
t = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
gathered = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(gathered, t)
out = X.combine("eip2537_hip_pairing_combine", [bytes(g.numpy().tobytes()) for g in gathered])
assert out == bytes(31) + b"\x01", "sharded pairing check failed"
pairs[0:128] = m.encode_g1(m.g1_mul(m.G1, 12345))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_bench_self_launch_runs_the_ranks(tmp_path, clib, X):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (bench.spawn_ranks):
    the same launcher drives the gloo worker above -- rendezvous on 127.0.0.1, RANK / LOCAL_RANK /
    WORLD_SIZE per child, worst exit status returned -- and a failing rank fails the launch."""
    import bench
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    assert bench.spawn_ranks(2, [ROOT], script=str(script)) == 0
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(30)\n")
    assert bench.spawn_ranks(2, [], script=str(bad)) == 3          # rank 0 is stopped, status of the failed rank


def test_bench_gpus_flag_is_not_ignored():
    """--gpus N must either run N ranks or fail: here (no GPU) every rank refuses to start, and
    a mismatching WORLD_SIZE is refused instead of being reported as n_gpus = 1."""
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert cp.returncode != 0 and "n_gpus" not in cp.stdout
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert cp.returncode == 2 and "refusing" in cp.stderr and "n_gpus" not in cp.stdout


def _gloo_ranks(tmp_path, world, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "ok" in o


def test_two_rank_gloo_shard_gather_combine(tmp_path, clib, X):
    _gloo_ranks(tmp_path, 2, 29611)


def test_eight_rank_gloo_shard_gather_combine(tmp_path, clib, X):
    """BASELINE configs[4] is 8 GPUs: the same exchange with eight ranks (75-record shards, one pair per rank)."""
    _gloo_ranks(tmp_path, 8, 29612)
