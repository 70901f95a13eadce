"""ThreadSanitizer pass over the library's host concurrency code (the reference runs `go test -race`; this is
the engine's own counterpart).  csrc/api.hip -- engine slot pool (SlotLease), coalescing queue (Batcher / coalesce),
record-range split over devices (run_shards) and the pipelined shards of one device (CopyGate), last-call statistics -- is compiled for the host only against stand-ins for
the HIP runtime and the device pipelines (tools/tsan/) and hammered from 24 threads: every result must equal the one the
same call returns on an idle library and ThreadSanitizer must stay silent.  No GPU code is involved; the sanitizer build is
never run on a GPU box (not a `gpu` test)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not (os.path.exists(CLANG) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))),
                    reason="ROCm clang / hipcc not installed")
def test_host_concurrency_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "eip_tsan_hammer")
    b = subprocess.run([os.path.join(ROOT, "tools", "tsan", "build.sh"), exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert b.returncode == 0, b.stdout[-3000:]
    env = dict(os.environ, EIP2537_HIP_SLOTS="2", EIP2537_HIP_SPLIT_MIN="1024", EIP_STUB_NDEV="2",
               TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("EIP2537_HIP_DEVICES", None)
    env.pop("EIP2537_HIP_DEVICE", None)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " 0 mismatches" in r.stdout and "coalesced" in r.stdout, r.stdout
    # ONE stub device: the large G1 / G2 cases run as three shards on engine slots of the same device whose copies follow each
    # other (api.hip: pipeline_shards, CopyGate) -- with only two slots, so that shards wait for slots while others hold them
    env.update(EIP_STUB_NDEV="1", EIP2537_H2D_PIPELINE="3")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " 0 mismatches" in r.stdout, r.stdout
    # staged G1 calls (round 4: StagedCopy / Helper of csrc/engine.h -- the slot's helper thread copies the record shards while the
    # calling thread consumes them), on one stub device and nested inside the split over two
    for ndev in ("1", "2"):
        env.update(EIP_STUB_NDEV=ndev, EIP2537_H2D_PIPELINE="1", EIP2537_H2D_STAGES="1,3,3,2")
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert " 0 mismatches" in r.stdout, r.stdout
