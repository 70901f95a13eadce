"""bench.py on the GPU box: the one-GPU line carries both halves of BASELINE's metric with their rooflines, and the
N > 1 line (rehearsed with two gloo ranks sharing the one GPU of the test box: everything but RCCL itself) carries the
weak-scaling headline, the strong-scaling legs of BASELINE config 5 and the in-library split."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, env=None, timeout=900):
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_reports_both_halves():
    d = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-host-abi"])
    assert d["n_gpus"] == 1 and d["bit_exact_vs_golden"] is True
    assert d["roofline"]["frac"] > 0 and d["roofline_valu"]["frac"] > 0
    sec = d["secondary"]
    assert sec["result_is_one"] is True
    assert sec["roofline"]["kernel"].startswith("k_pair_lines8")
    kernels = {k["kernel"]: k for k in sec["roofline_valu"]["kernels"]}
    assert set(kernels) == {"k_pair_lines8", "k_pair_check_g1", "k_pair_fold"}
    for k in kernels.values():
        assert k["ms"] > 0 and 0 < k["frac"] < 1


def test_two_rank_line_has_weak_strong_and_split():
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "18"], env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["records_total"] == 2 << 18
    st = d["strong"]
    assert st["g1msm"]["records_total"] == 1 << 20 and st["g1msm"]["bit_exact_vs_golden"] is True
    assert st["pairing"]["records_total"] == 1 << 12 and st["pairing"]["bit_exact_vs_golden"] is True
    assert "in_library_split" in d and "error" not in d["in_library_split"]


def test_single_rank_through_rccl():
    """The N > 1 step -- per-rank partial, pinned host -> device copy, RCCL all_gather_into_tensor, device -> host copy, combine -- with the
    REAL collective library: one rank, backend nccl (= RCCL), BENCH_FORCE_DIST=1.  Several ranks cannot share the one GPU of the test box
    under RCCL, so this is as close as a one-GPU box gets to the driver's multi-GPU run."""
    env = dict(os.environ, BENCH_FORCE_DIST="1")
    env.pop("BENCH_DIST_BACKEND", None)
    d = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-host-abi", "--no-secondary", "--sustained", "0"], env=env)
    assert d["n_gpus"] == 1 and d["bit_exact_vs_golden"] is True
    assert "all_gather" in d["config"]["workload"], d["config"]
