"""CPU tests of the drop-in boundary (no GPU, no compute on the device): the library loads,
exports every symbol include/*.h declares, and its host-side single operations and length
rules match the oracle and the golden KATs."""
import ctypes
import json
import os
import re
import subprocess

import bls12_381 as m
from conftest import ROOT, call_x

GOLD = os.path.join(ROOT, "tests", "golden")


def _declared_symbols():
    funcs, objs = set(), set()
    for h in ("eip2537.h", "eip2537_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        funcs |= set(re.findall(r"\b((?:bls12|eip2537_hip)_\w+)\s*\(", src))
        objs |= set(re.findall(r"extern const uint64_t (\w+)", src))
    return funcs, objs


def test_library_exports_every_declared_symbol(X):
    import blst_eip2537_amd as pkg
    L = ctypes.CDLL(pkg.lib_path())
    funcs, objs = _declared_symbols()
    assert len(funcs) >= 13 + 9 + 10 and len(objs) == 11
    for name in sorted(funcs | objs):
        assert hasattr(L, name), "missing export: " + name
    # the 13 names the Rust extern block / Go wrappers bind (rust/src/lib.rs:18-96)
    for name in ["bls12_g1add", "bls12_g1mul", "bls12_g1multiexp", "bls12_g1multiexp_naive", "bls12_g1multiexp_bc",
                 "bls12_g2add", "bls12_g2mul", "bls12_g2multiexp", "bls12_g2multiexp_naive", "bls12_g2multiexp_bc",
                 "bls12_pairing", "bls12_map_fp_to_g1", "bls12_map_fp2_to_g2"]:
        assert name in funcs
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.lib_path()], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert not any(s.startswith("oracle_") for s in exported), "product must not contain oracle symbols"


def test_gas_constants_and_functions(X):
    import blst_eip2537_amd as pkg
    L = ctypes.CDLL(pkg.lib_path())
    val = lambda n: ctypes.c_uint64.in_dll(L, n).value
    assert (val("BLS12_G1ADD_GAS"), val("BLS12_G1MUL_GAS"), val("BLS12_G2ADD_GAS"), val("BLS12_G2MUL_GAS")) == (600, 12000, 4500, 55000)
    assert (val("BLS12_PAIRING_BASE_GAS"), val("BLS12_PAIRING_PAIR_GAS")) == (115000, 23000)
    assert (val("BLS12_MAP_FP_TO_G1_GAS"), val("BLS12_MAP_FP2_TO_G2_GAS")) == (5500, 110000)
    tab = (ctypes.c_uint64 * 128).in_dll(L, "BLS12_MULTIEXP_DISCOUNT")
    assert tab[0] == 1200 and tab[127] == 174 and val("BLS12_MULTIEXP_DISCOUNT_TABLE_LEN") == 128
    assert X.gas("g1add") == 600 and X.gas("map_fp2_to_g2") == 110000
    for k in [0, 1, 2, 127, 128, 129, 4096]:
        want = 0 if k == 0 else k * 12000 * tab[min(k, 128) - 1] // 1000
        assert X.gas("g1multiexp", 160 * k) == want
        want2 = 0 if k == 0 else k * 55000 * tab[min(k, 128) - 1] // 1000
        assert X.gas("g2multiexp", 288 * k + 5) == want2
        assert X.gas("pairing", 384 * k) == (115000 + 23000 * k if k else 0)


def test_host_single_ops_match_kat(X):
    with open(os.path.join(GOLD, "kat.json")) as f:
        kat = json.load(f)
    fn = {"g1add": X.g1_add, "g1mul": X.g1_mul, "g2add": X.g2_add, "g2mul": X.g2_mul,
          "map_fp_to_g1": X.map_fp_to_g1, "map_fp2_to_g2": X.map_fp2_to_g2}
    n = 0
    for v in kat:
        if v["op"] in fn:
            want = (v["code"], bytes.fromhex(v["output"]) if v["output"] is not None else None)
            assert call_x(fn[v["op"]], bytes.fromhex(v["input"])) == want, v["op"]
            n += 1
    assert n >= 40


def test_host_single_ops_match_oracle_random(X, clib):
    rng = m.SplitMix64(2024)
    for i in range(6):
        p, q = m.random_g1(rng, i % 2 == 0), m.random_g1(rng, True)
        inp = m.encode_g1(p) + m.encode_g1(q)
        assert call_x(X.g1_add, inp) == clib.call("bls12_g1add", inp)
        inp = m.encode_g1(p) + m.encode_scalar(rng.scalar256())
        assert call_x(X.g1_mul, inp) == clib.call("bls12_g1mul", inp)
    for i in range(3):
        p, q = m.random_g2(rng, i % 2 == 0), m.random_g2(rng, True)
        inp = m.encode_g2(p) + m.encode_g2(q)
        assert call_x(X.g2_add, inp) == clib.call("bls12_g2add", inp)
        inp = m.encode_g2(p) + m.encode_scalar(rng.scalar256())
        assert call_x(X.g2_mul, inp) == clib.call("bls12_g2mul", inp)


def test_length_rules_need_no_device(X):
    """INVALID_LENGTH is decided before the input is touched or the device is initialised
    (reference src/eip2537.c:436,489,543,724,777,831,1022,1097,1139)."""
    for fn, bad in [(X.g1_add, 255), (X.g1_mul, 159), (X.g2_add, 513), (X.g2_mul, 0), (X.g1_multiexp, 0),
                    (X.g1_multiexp, 161), (X.g1_multiexp_naive, 1), (X.g1_multiexp_bc, 319), (X.g2_multiexp, 0),
                    (X.g2_multiexp, 289), (X.g2_multiexp_naive, 287), (X.g2_multiexp_bc, 1), (X.pairing, 0),
                    (X.pairing, 385), (X.map_fp_to_g1, 63), (X.map_fp2_to_g2, 129)]:
        assert call_x(fn, bytes(bad)) == (5, None)
    # invalid field element for the map precompiles
    assert call_x(X.map_fp_to_g1, bytes(15) + b"\x01" + bytes(48))[0] == 3
    assert call_x(X.map_fp2_to_g2, m.encode_fp(1) + bytes(16) + m.P.to_bytes(48, "big"))[0] == 3


def test_generators_agree_with_oracle(X, clib):
    a, b = 0x1f3a5c7e9b2d4f60, 0x0123456789abcdef0fedcba987654321
    assert X.gen_msm_input("g1", 50, a, b, 42) == clib.gen_msm_input("g1", 50, a, b, 42)
    assert X.gen_msm_input("g2", 20, a, b, 42) == clib.gen_msm_input("g2", 20, a, b, 42)
    full = X.gen_msm_input("g1", 50, a, b, 42)
    assert X.gen_msm_input("g1", 20, a, b, 42, start=30) == full[30 * 160:]
    assert X.gen_pairing_input(10, a, b, b, a) == clib.gen_pairing_input(10, a, b, b, a)


def test_no_cpu_fallback_without_device(X):
    """On a box without a HIP device the hot path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    inp = m.encode_g1(m.G1) + m.encode_scalar(5)
    assert call_x(X.g1_multiexp, inp * 3) == (7, None)
    assert call_x(X.pairing, m.encode_g1(m.G1) + m.encode_g2(m.G2)) == (7, None)


def test_python_layer_keeps_one_hip_runtime_in_the_process():
    """PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both open the device (whichever
    comes second reports "no ROCm-capable device").  The Python layer maps PyTorch's copy before the engine library when torch
    is installed but not imported yet, so the order of `import torch` and the first engine call does not matter."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import blst_eip2537_amd as p\np.lib()\nimport torch\n"
            "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
            "print('HIPLIBS', len(libs), libs)\n" % ROOT)
    cp = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert "HIPLIBS 1 " in cp.stdout, cp.stdout[-2000:]


def test_bound_hip_runtime_is_reported():
    """the Python layer maps PyTorch's bundled HIP runtime before the engine (one runtime per process); which copy
    was bound must be visible, and the preload must be something a caller can turn off."""
    import blst_eip2537_amd as pkg
    assert isinstance(pkg.bound_hip_runtime(), str) and pkg.bound_hip_runtime()
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['EIP2537_HIP_PRELOAD_TORCH_RUNTIME'] = '0'\n"
            "import blst_eip2537_amd as pkg; print(pkg.bound_hip_runtime())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    assert "preload disabled" in out, out
