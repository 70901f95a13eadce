#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native EIP-2537 engine.

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: by default one bls12_g1multiexp over 2^20 (point, scalar) records
(BASELINE.json metric "G1 MSM pairs/sec at 2^20"), called through the C-ABI
(eip2537_hip_g1multiexp_dev).

`--gpus N` with N > 1 runs N ranks, one process per GPU (torch.distributed, backend nccl = RCCL).
Launched by torchrun / torch.distributed.run the ranks are already there (WORLD_SIZE is set);
launched as plain `python bench.py --gpus N` this process starts them itself -- before it imports
torch or touches HIP -- and exits with their status.  ONE larger MSM is sharded by contiguous
record range, 2^20 records per GPU (weak scaling: the metric's size per GPU): every rank reduces
its shard to one 192-byte partial point, the partials are all-gathered over RCCL and combined into
the precompile's output.  `--scaling strong` instead shards one 2^20-record MSM over the N GPUs
(BASELINE config 5 literally; at 2^17 records per GPU the fixed chain latency dominates).

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline       dominant kernel (named by the library: eip2537_hip_last_plan) against the HBM roof,
                 from HIP events recorded on the engine's own stream around that kernel
  roofline_valu  the same kernel against the v_mad_u64_u32 issue roof it actually lives under
  host_abi       the reference-ABI call itself, bls12_g1multiexp on a host buffer (H2D included:
                 SURVEY.md 8d's definition of the metric; never `value`)
  cpu_baseline   the CPU oracle's restatement of the reference path (Bos-Coster, 1 thread) on the
                 same records, on this box's host cores; cpu_all_cores: a bucket-method MSM on all
                 host cores, for context, NOT the reference's algorithm
  sustained      >= 2 s of back-to-back device-resident steps (the timed region above is a 0.07 s burst, and the part lowers its
                 clock under sustained VALU load): mean / median ms per step and the ratio to the burst; the same for the pairing check
  secondary      the second half of BASELINE's metric: one 2^12-pair bls12_pairing check, with its own roofline
                 (HBM fraction + PMC traffic of the line walk), roofline_valu (walk, G1 membership, line products) and cpu_baseline
  strong         (N > 1) BASELINE config 5 literally: ONE 2^20-record MSM and ONE 2^12-pair check sharded over the N GPUs
  in_library_split  (N > 1) bls12_g1multiexp on a host buffer, cut over the N devices inside the library
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# synthetic workload parameters: P_i = [A + i*B]G, k_i = SplitMix64(SEED)
A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321
R_ORDER = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MAD_PEAK = 33775e9             # v_mad_u64_u32 lane-ops/s, chip-wide, SUSTAINED (tools/valu_probe.hip 2.0: 33.8 T from the second 34-ms window on,
                               # profiles/r04_valu_probe.txt).  Rounds 1-3 priced against 29.44 T, the probe's first 3-ms burst, taken before the
                               # clock has ramped up: every VALU fraction of this round is therefore ~13 % LOWER than the same kernel's in round 3
MADS_PER_FP_PRODUCT = 338      # 13 x 13 product + 13 x 13 reduction columns of 30-bit limbs (csrc/field.h)
REC = {"g1msm": 160, "g2msm": 288, "pairing": 384}          # algorithmic bytes per unit
FULL = {"g1msm": "eip2537_hip_g1multiexp_dev", "g2msm": "eip2537_hip_g2multiexp_dev",
        "pairing": "eip2537_hip_pairing_dev"}
PART = {"g1msm": "eip2537_hip_g1msm_partial_dev", "g2msm": "eip2537_hip_g2msm_partial_dev",
        "pairing": "eip2537_hip_pairing_partial_dev"}
COMB = {"g1msm": "eip2537_hip_g1msm_combine", "g2msm": "eip2537_hip_g2msm_combine",
        "pairing": "eip2537_hip_pairing_combine"}
ORACLE = {"g1msm": "bls12_g1multiexp", "g2msm": "bls12_g2multiexp", "pairing": "bls12_pairing"}
HOSTFN = {"g1msm": "g1_multiexp", "g2msm": "g2_multiexp", "pairing": "pairing"}


def seed_for(workload, log2n):
    return 0x25370000 + {"g1msm": 0, "g2msm": 0x100, "pairing": 0x200}[workload] + log2n


def make_records(X, workload, n, start, log2n_total):
    if workload == "pairing":
        return X.gen_pairing_input(n, A, B, B ^ 0x55, A ^ 0x33, start)
    return X.gen_msm_input("g1" if workload == "g1msm" else "g2", n, A, B, seed_for(workload, log2n_total), start)


def pairing_fixup(X, buf, n_total):
    """Replace the last pair by ([c]G1, G2), c = -sum a_i b_i, so that the product is one."""
    a0, a1, b0, b1 = A, B, B ^ 0x55, A ^ 0x33
    s = 0
    for i in range(n_total - 1):
        s += ((a0 + i * a1) % R_ORDER) * ((b0 + i * b1) % R_ORDER)
    c = (-s) % R_ORDER
    gen = X.gen_pairing_input(1, 1, 0, 1, 0)           # (G1, G2)
    last = X.g1_mul(gen[:128] + c.to_bytes(32, "big")) + gen[128:]
    return buf[:-384] + last


def golden(workload, log2n):
    p = os.path.join(ROOT, "tests", "golden", "%s_2p%d.hex" % (workload, log2n))
    if os.path.exists(p):
        with open(p) as f:
            return bytes.fromhex(f.read().strip())
    return None


# multiply-adds the pairing kernels execute per pair (csrc/pairing_limb.h): a round is one two-product sum with a single
# reduction (507) per lane of the 8-lane walk, one product (338) per lane of the 4-lane G1 membership chain; a folded line
# costs a quad of lanes 9 two-product sums + 3 products each, once per Miller step
PAIR_MADS = {
    "k_pair_lines8": (63 * 2 + 5 * 4 + 2) * 507 * 8,          # 63 doublings x 2 rounds, 5 additions x 4, membership 2
    "k_pair_check_g1": (126 * 2 + 10 * 3 + 2) * 338 * 4,      # two 63-doubling chains x 2 rounds, 10 additions x 3, final 2
    "k_pair_fold": 68 * 4 * (9 * 507 + 3 * 338),             # 68 lines per pair
}
# round 4: checks of up to 2^13 pairs fold their lines with three six-product sums (one reduction each: 6 x 169 + 169) instead of nine
# two-product sums -- fewer multiply-adds executed for the same lines (csrc/pairing_limb.h, quad_fold_line<true>)
PAIR_FOLD_FUSED = 68 * 4 * (3 * (6 * 169 + 169) + 3 * 338)


def pairing_valu(k, walk_ms, check_ms, fold_ms, pipeline_ms):
    """Executed v_mad_u64_u32 lane-ops of the three pairing kernels against the measured issue roof."""
    def leg(name, ms):
        mads = (PAIR_FOLD_FUSED if name == "k_pair_fold" and k <= 8192 else PAIR_MADS[name]) * k
        return {"kernel": name, "ms": ms, "lane_mads": mads, "achieved": mads / (ms * 1e-3) / 1e12 if ms > 0 else None,
                "frac": mads / (ms * 1e-3) / MAD_PEAK if ms > 0 else None}
    total = (sum(PAIR_MADS.values()) - (PAIR_MADS["k_pair_fold"] - PAIR_FOLD_FUSED if k <= 8192 else 0)) * k
    return {"bound": "valu (v_mad_u64_u32 issue)", "peak": MAD_PEAK / 1e12, "unit": "T mad lane-ops/s",
            "kernels": [leg("k_pair_lines8", walk_ms), leg("k_pair_check_g1", check_ms), leg("k_pair_fold", fold_ms)],
            "device_pipeline": {"lane_mads": total, "ms": pipeline_ms, "frac": total / (pipeline_ms * 1e-3) / MAD_PEAK if pipeline_ms > 0 else None},
            "note": "executed multiply-adds per pair x pairs / kernel time from HIP events on the engine's streams (membership runs on a second "
                    "stream beside the walk; k_pair_fold's interval includes k_pair_tree2); peak = chip-wide v_mad_u64_u32 rate measured by "
                    "tools/valu_probe.hip, sustained form (profiles/r04_valu_probe.txt)"}


def traffic_for(wl, log2n, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc summary of this same command
    (profiles/rNN_traffic.json, newest first; counters cannot be read from inside the process)."""
    for tname in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tpath):
            continue
        with open(tpath) as f:
            tj = json.load(f)
        for ent in (tj if isinstance(tj, list) else [tj]):
            if ent.get("workload") == wl and ent.get("log2n") == log2n and ent.get("kernel") == kernel:
                note = ("PMC (FETCH_SIZE+WRITE_SIZE)*1024 per launch, raw; %.3g with the gfx950 x2 FETCH correction (calibrated for coalesced "
                        "streams only). Source: profiles/%s" % (ent["traffic_bytes_fetch_x2"], ent.get("source", tname)))
                if ent.get("note"):
                    note += "; " + ent["note"]
                return ent["traffic_bytes_raw"], note
    return None, None


def stats_ms(samples):
    return {"mean": sum(samples) / len(samples), "median": statistics.median(samples), "min": min(samples),
            "max": max(samples), "n": len(samples)}


class stdout_to_stderr:
    """Backends print connection chatter ("[Gloo] Rank 0 is connected to ...") to the C-level stdout; the
    driver expects ONE JSON line there, so rendezvous runs with fd 1 pointed at stderr."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, env=None, script=None):
    """Start n copies of this script, one rank each, and return the worst exit status.  Called before
    anything in this process has imported torch or initialised HIP, and the children are started as
    child processes (never exec'd over a process that holds the GPU)."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base["WORLD_SIZE"] = str(n)
    base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=e))
    worst = 0
    deadline = time.time() + float(os.environ.get("BENCH_SPAWN_TIMEOUT", "1500"))
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    worst = worst or rc
        if worst or time.time() > deadline:
            for r in pending:                    # a rank failed (or the run hung): stop exactly our children
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            return worst or 124
        time.sleep(0.05)
    return worst


def sustained_leg(X, fn_name, d_ptr, n, seconds, burst_ms):
    """Back-to-back device-resident calls for at least `seconds` (no pause between calls beyond the call's own host tail)."""
    ts = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(ts) < 10:
        t0 = time.perf_counter()
        X.dev_call(fn_name, d_ptr, n)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = stats_ms(ts)
    q = max(1, len(ts) // 4)
    first, last = sum(ts[:q]) / q, sum(ts[-q:]) / q
    return {"seconds": sum(ts) / 1e3, "steps": len(ts), "ms_per_step": st, "value": n / (st["mean"] * 1e-3), "unit": "pairs/s",
            "first_quarter_ms": first, "last_quarter_ms": last, "burst_ms_per_step": burst_ms,
            "sustained_over_burst": st["mean"] / burst_ms if burst_ms else None,
            "note": "same call as the timed steps, repeated back to back for >= %.1f s; sustained_over_burst > 1 means slower under sustained load" % seconds}


def host_abi_leg(X, wl, host, n, reps):
    """The reference-ABI call on a host buffer: H2D + decode + compute + encode."""
    fn = getattr(X, HOSTFN[wl])
    for _ in range(3):                            # warm-up: grows the staging buffers of every engine slot a pipelined call may land on
        out = fn(host)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn(host)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = stats_ms(ts)
    return out, {"call": "bls12_%s (host buffer, %d bytes: H2D + decode + compute + encode)" % (ORACLE[wl][6:], len(host)),
                 "ms_per_call": st, "value": n / (st["median"] * 1e-3), "value_best": n / (st["min"] * 1e-3),
                 "unit": "pairs/s", "devices": X.device_count(), "plan": X.last_plan()}


def split_child(args):
    """`--split-child D`: a fresh process (no torch, no RCCL) that lets the LIBRARY cut one host-input
    bls12_g1multiexp / bls12_pairing over the devices in $EIP2537_HIP_DEVICES and prints one JSON object."""
    import blst_eip2537_amd as pkg
    X = pkg.Eip2537Executor
    wl = args.workload
    log2n = args.log2n if args.log2n is not None else {"g1msm": 20, "g2msm": 16, "pairing": 12}[wl]
    n = 1 << log2n
    host = make_records(X, wl, n, 0, log2n)
    if wl == "pairing":
        host = pairing_fixup(X, host, n)
    out, leg = host_abi_leg(X, wl, host, n, max(3, args.steps))
    gold = bytes(31) + b"\x01" if wl == "pairing" else golden(wl, log2n)
    leg["bit_exact_vs_golden"] = None if gold is None else (out == gold)
    leg["EIP2537_HIP_DEVICES"] = os.environ.get("EIP2537_HIP_DEVICES")
    print(json.dumps(leg), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["g1msm", "g2msm", "pairing"], default="g1msm")
    ap.add_argument("--log2n", type=int, default=None, help="log2 of the batch (default 20 / 16 / 12)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak")
    ap.add_argument("--window", type=int, default=0, help="force the Pippenger window width (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-host-abi", action="store_true")
    ap.add_argument("--sustained", type=float, default=2.0, help="seconds of back-to-back steps for the `sustained` leg (0: skip)")
    ap.add_argument("--split-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.split_child:
        return split_child(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (nothing here has touched HIP yet)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a wrong n_gpus\n" % (args.gpus, world))
        sys.exit(2)

    import torch
    import blst_eip2537_amd as pkg
    X = pkg.Eip2537Executor

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the multiexp / pairing path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a GPU to rehearse the
    # sharded path on a 1-GPU box (everything but RCCL itself)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s); RCCL needs one GPU per rank "
                         "(BENCH_DIST_BACKEND=gloo rehearses the sharded path on fewer)" % (world, ndev))
    dev_index = local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dist = None
    # BENCH_FORCE_DIST=1: a single rank still goes through the process group, the partial / all_gather / combine step and RCCL itself
    # (the only way to exercise the N > 1 code path with the real collective library on a one-GPU box)
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    X.init(dev_index)                     # one process per GPU: pin the library to this rank's device
    if args.window:
        X.set_window(args.window)

    wl = args.workload
    log2n = args.log2n if args.log2n is not None else {"g1msm": 20, "g2msm": 16, "pairing": 12}[wl]
    if args.scaling == "strong":
        n_total = 1 << log2n
        n_local = n_total // world
    else:
        n_local = 1 << log2n
        n_total = n_local * world
    start = rank * n_local
    log2_total = log2n if args.scaling == "strong" else log2n + (world.bit_length() - 1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(lwl, ln_local, ln_total, lstart, llog2_total, steps, warmup):
        """W warm-up + K timed steps of one workload, sharded by contiguous record range over the ranks: every rank reduces
        its shard to one partial, the partials cross the ranks in ONE all_gather (RCCL) and every rank combines them.  The
        partial leaves the library on the host (the window Horner / the 63-squaring Horner are host code), so a step is:
        pinned host -> device copy, all_gather_into_tensor, one device -> host copy, one stream synchronisation."""
        lhost = make_records(X, lwl, ln_local, lstart, llog2_total)
        if lwl == "pairing" and rank == world - 1:
            lhost = pairing_fixup(X, lhost, ln_total)
        d_buf = torch.frombuffer(bytearray(lhost), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        lpsz = {"g1msm": 192, "g2msm": 384, "pairing": 576}[lwl]
        pin_in = pin_out = dev_in = gbuf = None
        if (world > 1 or force_dist) and backend == "nccl":
            pin_in = torch.empty(lpsz, dtype=torch.uint8).pin_memory()
            pin_out = torch.empty(world * lpsz, dtype=torch.uint8).pin_memory()
            dev_in = torch.empty(lpsz, dtype=torch.uint8, device="cuda")
            gbuf = torch.empty(world * lpsz, dtype=torch.uint8, device="cuda")
        k_ms, p_ms, s_ms, aux = [], [], [], []

        def one():
            t0 = time.perf_counter()
            if world == 1 and not force_dist:
                o = X.dev_call(FULL[lwl], d_buf.data_ptr(), ln_local)
            else:
                part = X.dev_call(PART[lwl], d_buf.data_ptr(), ln_local)
                if backend == "nccl":
                    pin_in.copy_(torch.frombuffer(bytearray(part), dtype=torch.uint8))
                    dev_in.copy_(pin_in, non_blocking=True)
                    dist.all_gather_into_tensor(gbuf, dev_in)
                    pin_out.copy_(gbuf, non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                    allp = bytes(pin_out.numpy().tobytes())
                else:
                    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
                    parts = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(parts, mine)
                    allp = b"".join(bytes(t.numpy().tobytes()) for t in parts)
                o = X.combine(COMB[lwl], [allp[i * lpsz:(i + 1) * lpsz] for i in range(world)])
            s_ms.append((time.perf_counter() - t0) * 1e3)
            pm, km = X.last_timing()
            p_ms.append(pm)
            k_ms.append(km)
            aux.append(X.last_timing_aux())
            return o

        o = None
        for _ in range(warmup):
            o = one()
        k_ms.clear(); p_ms.clear(); s_ms.clear(); aux.clear()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = one()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return {"elapsed": el, "out": o, "kernel_ms": k_ms, "pipe_ms": p_ms, "step_ms": s_ms, "aux_ms": aux, "host": lhost,
                "plan": X.last_plan() or {}, "psz": lpsz}

    leg0 = timed_leg(wl, n_local, n_total, start, log2_total, args.steps, args.warmup)
    host, out, plan, psz = leg0["host"], leg0["out"], leg0["plan"], leg0["psz"]
    kernel_ms, pipe_ms, step_ms, elapsed = leg0["kernel_ms"], leg0["pipe_ms"], leg0["step_ms"], leg0["elapsed"]

    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    value = n_total / (ms_per_step * 1e-3)

    # correctness of what was timed: analytic golden fixture (if one exists for this size)
    gold = golden(wl, log2_total)
    if wl == "pairing":
        gold = bytes(31) + b"\x01"
    parity = None if gold is None else (out == gold)

    # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this
    # same command; the summary is committed under profiles/ and only quoted when it was taken on
    # the workload and kernel being run.
    traffic, traffic_note = (None, None) if world > 1 else traffic_for(wl, log2n, (plan.get("kernel") or "").split("<")[0] if wl == "pairing" else plan.get("kernel"))

    result = None
    if rank == 0:
        k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        p_ms = sum(pipe_ms) / max(1, len(pipe_ms))
        achieved = REC[wl] * n_local / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        st = stats_ms(step_ms)
        result = {
            "metric": {"g1msm": "g1_msm_pairs_per_sec", "g2msm": "g2_msm_pairs_per_sec",
                       "pairing": "pairing_pairs_per_sec"}[wl],
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Montgomery integer arithmetic)",
            "data": "synthetic",
            "config": {"workload": "%s over 2^%d records total (%d per GPU), input resident in HBM, via C-ABI %s"
                                   % (ORACLE[wl], log2_total, n_local, FULL[wl] if world == 1 and not force_dist else PART[wl] + " + RCCL all_gather + combine"),
                       "records_total": n_total, "records_per_gpu": n_local, "world_size": world,
                       "parallelism": "1 GPU" if world == 1 else "record-range shards x%d, RCCL all_gather of %d-byte partials" % (world, psz)},
            "step_ms": st,                                     # rank 0's own clock around every step
            "value_median": n_total / (st["median"] * 1e-3), "value_best": n_total / (st["min"] * 1e-3),
            "bit_exact_vs_golden": parity,
            # the reference Go bench's own unit (go/blst_eip2537_test.go:126-130), gas of the whole input
            "mgas_per_s": X.gas({"g1msm": "g1multiexp", "g2msm": "g2multiexp", "pairing": "pairing"}[wl],
                                n_total * REC[wl]) / (ms_per_step * 1e-3) / 1e6,
            "plan": plan,
            "roofline": {"bound": "hbm", "kernel": plan.get("kernel"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel_ms": k_ms, "device_pipeline_ms": p_ms,
                         "algorithmic_bytes_per_unit": REC[wl], "units_per_launch": n_local,
                         "note": "integer-VALU-bound by construction (SURVEY.md 8d): the HBM fraction is reported because the metric asks for it"},
        }
        # the roof this path actually lives under: multiply-adds issued per second by the dominant
        # kernel against the measured chip-wide v_mad_u64_u32 issue rate
        if wl in ("g1msm", "g2msm") and k_ms > 0 and plan.get("windows"):
            prods = n_local * plan["windows"] * (10 if wl == "g1msm" else 30)      # 8M+2S per mixed addition; Fp2 = 3 Fp
            mads = prods * MADS_PER_FP_PRODUCT
            result["roofline_valu"] = {
                "bound": "valu (v_mad_u64_u32 issue)", "kernel": plan.get("kernel"),
                "achieved": mads / (k_ms * 1e-3) / 1e12, "peak": MAD_PEAK / 1e12, "unit": "T mad lane-ops/s",
                "frac": mads / (k_ms * 1e-3) / MAD_PEAK, "fp_products_per_launch": prods,
                "note": "algorithmic Fp products = records x windows (%d, c = %d) x 10 (x3 over Fp2), %d multiply-adds each; "
                        "peak = chip-wide v_mad_u64_u32 rate measured by tools/valu_probe.hip, sustained form (profiles/r04_valu_probe.txt)"
                        % (plan["windows"], plan.get("window_bits", 0), MADS_PER_FP_PRODUCT)}
            # multiply-adds the G1 kernels really execute per mixed addition: the two squarings take 260 of them
            # (91 + 169), and the limb-form kernel reduces R (Q - x3) - y1 PPP once (338 + 169)
            # G2 (two lanes per addition): k_msm_accum2c ten two-product sums per lane (507 each); k_msm_accum2c_l six of them, two
            # squares as one product each (338) and Y3 as one four-product sum (4 x 169 + 169) per lane
            executed = {"k_msm_accum_l": 6 * 338 + 2 * 260 + 507, "k_msm_accum<eip::Fp>": 8 * 338 + 2 * 260,
                        "k_msm_accum2<eip::Fp>": 8 * 338 + 2 * 260, "k_msm_accum2c": 2 * 10 * 507,
                        "k_msm_accum2c_l": 2 * (6 * 507 + 2 * 338 + 845)}.get(plan.get("kernel"))
            if executed:
                result["roofline_valu"]["mads_executed_per_addition"] = executed
                result["roofline_valu"]["frac_executed"] = n_local * plan["windows"] * executed / (k_ms * 1e-3) / MAD_PEAK

        if wl == "pairing" and k_ms > 0 and leg0["aux_ms"]:
            ck = sum(a[0] for a in leg0["aux_ms"]) / len(leg0["aux_ms"])
            fo = sum(a[1] for a in leg0["aux_ms"]) / len(leg0["aux_ms"])
            result["roofline_valu"] = pairing_valu(n_local, k_ms, ck, fo, p_ms)
        if wl in ("g1msm", "g2msm") and leg0["aux_ms"]:
            result["roofline"]["sort_stage_ms"] = sum(a[0] for a in leg0["aux_ms"]) / len(leg0["aux_ms"])
            result["roofline"]["fold_reduce_ms"] = sum(a[1] for a in leg0["aux_ms"]) / len(leg0["aux_ms"])

    # ---- the reference-ABI call itself: host buffer in, H2D inside the timed call
    if rank == 0 and world == 1 and not args.no_host_abi:
        hout, leg = host_abi_leg(X, wl, host, n_local, 12)
        leg["matches_device_resident_result"] = hout == out
        leg["note"] = ("SURVEY.md 8d's wording of the metric (the full call on a HOST buffer); `value` above stays the device-resident call because "
                       "this round's measurement rules fix it so (inputs resident in HBM when the timed region starts; the PCIe-inclusive rate "
                       "is reported, never `value`). `value` here = records / MEDIAN of %d calls." % leg["ms_per_call"]["n"])
        result["host_abi"] = leg

    # ---- sustained load: the timed region above is a burst (20 steps = 0.07 s); the same step back to back for >= 2 s
    if rank == 0 and world == 1 and args.sustained > 0:
        d_s = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        result["sustained"] = sustained_leg(X, FULL[wl], d_s.data_ptr(), n_local, args.sustained, ms_per_step)
        del d_s

    # ---- CPU baseline: oracle restatement of the reference path, 1 thread, on the same records
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import clib
        if wl == "pairing":
            sample_n = min(n_local, 2048)
            sample = pairing_fixup(X, host[:sample_n * 384], sample_n) if sample_n < n_local else host
        else:
            sample_n = min(n_local, 1 << 20 if wl == "g1msm" else 1 << 16)
            sample = host[:sample_n * REC[wl]]
        t1 = time.perf_counter()
        rc, cpu_out = clib.call(ORACLE[wl], sample)
        dt = time.perf_counter() - t1
        ok = rc == 0
        if sample_n == n_local:
            ok = ok and cpu_out == out
        else:
            d_s = torch.frombuffer(bytearray(sample), dtype=torch.uint8).cuda()
            ok = ok and X.dev_call(FULL[wl], d_s.data_ptr(), sample_n) == cpu_out
        result["cpu_baseline"] = {
            "value": sample_n / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%s 2^%d records of the same workload, oracle %s (reference control flow: Bos-Coster / sequential Miller loops), one run of %.1f s; host has %d cores (%d usable by this process)"
                      % ("all" if sample_n == n_local else "first", sample_n.bit_length() - 1, ORACLE[wl], dt, os.cpu_count() or 0,
                         len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0),
            "gpu_matches_cpu_on_sample": bool(ok),
        }
        if wl == "g1msm":
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(cores, 64)                 # the windows of the oracle's bucket method are the unit of work: ~20 of them
            t1 = time.perf_counter()
            rc, mt_out = clib.g1_pippenger_mt(sample, cores)
            dt = time.perf_counter() - t1
            result["cpu_all_cores"] = {
                "value": sample_n / dt, "unit": "pairs/s", "cores": cores, "kind": "NOT the reference's algorithm",
                "sample": "same 2^%d records, bucket-method MSM of the oracle on %d pthreads (windows dealt over threads), %.2f s: "
                          "context only -- the reference is single-threaded Bos-Coster (README.md:15-17)" % (sample_n.bit_length() - 1, cores, dt),
                "matches": rc == 0 and mt_out == cpu_out}

    # ---- secondary: the pairing half of the BASELINE metric (one 2^12-pair check)
    if rank == 0 and world == 1 and wl == "g1msm" and not args.no_secondary:
        k = 1 << 12
        ph = pairing_fixup(X, make_records(X, "pairing", k, 0, 12), k)
        d_p = torch.frombuffer(bytearray(ph), dtype=torch.uint8).cuda()
        pout = X.dev_call(FULL["pairing"], d_p.data_ptr(), k)
        torch.cuda.synchronize()
        reps, kms, pms, tms, cks, fos = 10, [], [], [], [], []
        for _ in range(reps):
            t1 = time.perf_counter()
            pout = X.dev_call(FULL["pairing"], d_p.data_ptr(), k)
            tms.append((time.perf_counter() - t1) * 1e3)
            pms.append(X.last_timing()[0])
            kms.append(X.last_timing()[1])
            cks.append(X.last_timing_aux()[0])
            fos.append(X.last_timing_aux()[1])
        pst = stats_ms(tms)
        pplan = X.last_plan() or {}
        ptraffic, ptraffic_note = traffic_for("pairing", 12, (pplan.get("kernel") or "").split("<")[0])
        sec = {"metric": "pairing_pairs_per_sec", "value": k / (pst["mean"] * 1e-3), "unit": "pairs/s", "ms_per_check": pst,
               "value_median": k / (pst["median"] * 1e-3), "value_best": k / (pst["min"] * 1e-3),
               "pairs": k, "result_is_one": pout == bytes(31) + b"\x01",
               "mgas_per_s": X.gas("pairing", k * 384) / (pst["mean"] * 1e-3) / 1e6,
               "roofline": {"bound": "hbm", "kernel": pplan.get("kernel"), "achieved": 384 * k / (sum(kms) / reps * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 384 * k / (sum(kms) / reps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": ptraffic, "traffic_note": ptraffic_note, "kernel_ms": sum(kms) / reps,
                            "device_pipeline_ms": sum(pms) / reps, "algorithmic_bytes_per_unit": 384, "units_per_launch": k},
               "roofline_valu": pairing_valu(k, sum(kms) / reps, sum(cks) / reps, sum(fos) / reps, sum(pms) / reps)}
        if args.sustained > 0:
            sec["sustained"] = sustained_leg(X, FULL["pairing"], d_p.data_ptr(), k, args.sustained, pst["mean"])
        if not args.no_host_abi:
            hout, leg = host_abi_leg(X, "pairing", ph, k, 12)
            leg["result_is_one"] = hout == bytes(31) + b"\x01"
            sec["host_abi"] = leg
        if not args.no_cpu_baseline:
            from oracle import clib
            t1 = time.perf_counter()
            rc, cout = clib.call("bls12_pairing", ph)
            dtc = time.perf_counter() - t1
            sec["cpu_baseline"] = {"value": k / dtc, "unit": "pairs/s", "cores": 1, "kind": "port",
                                   "sample": "the same 2^12-pair check, oracle bls12_pairing (reference control flow: sequential "
                                             "membership tests and Miller loops, one final exponentiation), one run of %.1f s" % dtc,
                                   "result_is_one": rc == 0 and cout == bytes(31) + b"\x01",
                                   "gpu_matches_cpu_on_sample": rc == 0 and cout == pout}
        result["secondary"] = sec

    # ---- N > 1: BASELINE config 5 literally -- ONE 2^20-record MSM and ONE 2^12-pair check sharded over the N GPUs
    # (strong scaling), next to the weak-scaling headline above.  Every rank runs the legs; rank 0 reports.
    if world > 1 and args.scaling == "weak" and not args.no_secondary:
        strong = {}
        for swl, slog in (("g1msm", 20), ("pairing", 12)):
            sn_total = 1 << slog
            if sn_total % world:
                continue
            sn_local = sn_total // world
            sl = timed_leg(swl, sn_local, sn_total, rank * sn_local, slog, max(3, min(args.steps, 10)), 2)
            nst = max(3, min(args.steps, 10))
            sgold = bytes(31) + b"\x01" if swl == "pairing" else golden(swl, slog)
            if rank == 0:
                sms = sl["elapsed"] * 1e3 / nst
                strong[swl] = {"metric": "%s_pairs_per_sec" % ("g1_msm" if swl == "g1msm" else "pairing"), "scaling": "strong",
                               "records_total": sn_total, "records_per_gpu": sn_local, "n_gpus": world, "steps": nst,
                               "ms_per_step": sms, "value": sn_total / (sms * 1e-3), "unit": "pairs/s",
                               "bit_exact_vs_golden": None if sgold is None else sl["out"] == sgold,
                               "shard_device_pipeline_ms": sum(sl["pipe_ms"]) / max(1, len(sl["pipe_ms"])),
                               "note": "one %s over 2^%d records cut into %d contiguous shards; a shard of this size is mostly fixed chain "
                                       "latency (DESIGN.md 6)" % (ORACLE[swl], slog, world)}
        if rank == 0:
            result["strong"] = strong

    # ---- N > 1: the same total input through the reference ABI of ONE process, cut over the N
    # devices inside the library (thread per device, host combine).  Runs in a fresh child of rank 0
    # while the other ranks wait on the host (a key in the rendezvous store: no collective spinning on
    # their GPUs), their GPUs idle.
    if world > 1 and wl in ("g1msm", "pairing") and not args.no_host_abi:
        if rank == 0:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
            env["EIP2537_HIP_DEVICES"] = ",".join(str(r % max(1, ndev)) for r in range(world))
            cmd = [sys.executable, os.path.abspath(__file__), "--split-child", "--workload", wl, "--steps", str(args.steps),
                   "--log2n", str(log2n if args.scaling == "strong" else min(log2_total, 23))]
            try:
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                result["in_library_split"] = json.loads(line[-1]) if cp.returncode == 0 and line else {"error": (cp.stderr or cp.stdout)[-400:]}
            except Exception as ex:                          # the headline number must survive this leg
                result["in_library_split"] = {"error": repr(ex)}
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("eip2537_split_leg_done", "1")
        else:
            import datetime
            try:
                store.wait(["eip2537_split_leg_done"], datetime.timedelta(seconds=900))     # longer than the child's own limit
            except Exception as ex:                                 # the headline number must survive this leg
                sys.stderr.write("bench.py rank %d: split leg wait: %r\n" % (rank, ex))

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
