"""The static forwarding shim (shim/libblst_eip2537.a): what the reference's untouched Rust
build script links (rust/build.rs:37-41).  No GPU needed: only host-side precompiles are called."""
import os
import subprocess

from conftest import ROOT

ABI13 = ["bls12_g1add", "bls12_g1mul", "bls12_g1multiexp", "bls12_g1multiexp_naive", "bls12_g1multiexp_bc",
         "bls12_g2add", "bls12_g2mul", "bls12_g2multiexp", "bls12_g2multiexp_naive", "bls12_g2multiexp_bc",
         "bls12_pairing", "bls12_map_fp_to_g1", "bls12_map_fp2_to_g2"]


def test_shim_archive_defines_abi_and_needs_only_libc(tmp_path, X):
    shim = os.path.join(ROOT, "shim")
    subprocess.check_call(["make", "-C", shim, "-s"])
    ar = os.path.join(shim, "libblst_eip2537.a")
    defined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "--defined-only", ar], text=True).splitlines()
               if len(ln.split()) >= 3}
    for s in ABI13 + ["BLS12_MULTIEXP_DISCOUNT", "bls12_pairing_gas", "BLS12_G1MUL_GAS"]:
        assert s in defined, s
    undefined = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-u", ar], text=True).splitlines() if ln.strip() and ":" not in ln}
    allowed = {"dlopen", "dlsym", "dlerror", "getenv", "fprintf", "__fprintf_chk", "stderr", "_GLOBAL_OFFSET_TABLE_"}
    assert undefined <= allowed, undefined - allowed      # libc only: no hip, no libstdc++
    # link a C client against the archive alone and run it against the engine .so, two threads
    import blst_eip2537_amd as pkg
    exe = str(tmp_path / "abi_client")
    subprocess.check_call(["gcc", "-O1", os.path.join(shim, "abi_client.c"), ar, "-lpthread", "-o", exe])
    env = dict(os.environ, EIP2537_HIP_LIB=pkg.lib_path())
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert "0 failures" in out.stdout
