"""Host run of the limb-form arithmetic behind k_msm_accum_l (csrc/limb30.h): the same HD source compiled
for the host by tools/limb30_check.hip, checked against the 64-bit host product and the generic madd()."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_limb_form_matches_host_arithmetic(tmp_path):
    exe = str(tmp_path / "limb30_check")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx",
                           "-I" + os.path.join(ROOT, "blst_eip2537_amd", "csrc"), os.path.join(ROOT, "tools", "limb30_check.hip"),
                           "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    for part in ("field operations", "accumulate chains", "point operations"):
        assert part + ": 0 mismatches" in out.stdout, out.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_pairing_lane_code_matches_host_arithmetic(tmp_path):
    """csrc/limbk.h + csrc/pairing_limb.h (bounded limb form, RCB projective doubling / addition, Miller walk,
    membership tests, quad line products, dense products through memory): the lane-group code of the pairing
    kernels with a group's shares in arrays, against pairing.h / curve.h (tools/pairing_limb_check.hip)."""
    exe = str(tmp_path / "pairing_limb_check")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx",
                           "-I" + os.path.join(ROOT, "blst_eip2537_amd", "csrc"), os.path.join(ROOT, "tools", "pairing_limb_check.hip"),
                           "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    # "G2 XYZZ on 8 lanes": csrc/g2_limb.h, the point operations of the G2 fold / bucket-reduce kernels, against curve.h
    for part in ("helpers", "G1 projective operations", "G1 membership", "Miller walk / G2 membership", "quad line products",
                 "dense products", "G2 XYZZ on 8 lanes"):
        assert part + ": 0 mismatches" in out.stdout, out.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_ifma_arithmetic_matches_scalar(tmp_path):
    """csrc/ifma.h (AVX-512 IFMA: eight Fp products per instruction, vector Fp12 products, cyclotomic squarings, the whole hard
    part of the final exponentiation and the Horner pass over a batch's per-group products) against the scalar code of
    pairing.h (tools/ifma_check.hip).  On a CPU without AVX-512 IFMA the tool reports that and the product uses the scalar code."""
    exe = str(tmp_path / "ifma_check")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-mbmi2", "-Xarch_host", "-madx",
                           "-I" + os.path.join(ROOT, "blst_eip2537_amd", "csrc"), os.path.join(ROOT, "tools", "ifma_check.hip"),
                           "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    if "skipped" in out.stdout:
        pytest.skip("no AVX-512 IFMA on this CPU")
    # "multiexp doubling chains": csrc/ifma_horner.h, the Horner doublings of a multiexp's host tail (G1 and G2) against curve.h
    for part in ("vector field operations", "cyclotomic squaring chains", "Fp12 products",
                 "exponentiation by z, final exponentiation, Horner", "multiexp doubling chains"):
        assert part + ": 0 mismatches" in out.stdout, out.stdout


def test_limb_form_column_accumulators_cannot_overflow():
    """tools/limb_column_bounds.py: exact worst case of every 64-bit column of mulL / sqrL / mul2L under the carry-out
    schedules of csrc/limb30.h (limbs < 2^30, the real limbs of p)."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "limb_column_bounds.py")], stdout=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout
