"""ASan + UBSan pass over the library's host code (codec, curve, pairing tower, map-to-curve, the small-call
host route: interleaved-window MSM and the shared-squaring Miller loop) -- tools/host_sanitize.sh.  GPU
sanitizers are not available on this pool; this is the part of the product that runs on the host."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None and shutil.which("hipcc") is None, reason="no host compiler")
def test_host_code_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["sh", os.path.join(ROOT, "tools", "host_sanitize.sh")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "host sanitize run: ok" in out.stdout
