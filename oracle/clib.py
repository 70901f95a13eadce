"""ctypes binding of the C oracle (oracle/liboracle_eip2537.so).  TEST INFRASTRUCTURE ONLY.

Every function returns (code, bytes|None) the way the reference's C-ABI reports results
(src/eip2537.h:31-59): code 0 and the fixed-size output, or a non-zero EIP2537_ERROR.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_eip2537.so")


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in _OUT_SIZES:
            f = getattr(_lib, "oracle_" + name)
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        for g in ("g1multiexp", "g2multiexp", "pairing"):
            f = getattr(_lib, "oracle_%s_gas" % g)
            f.restype = ctypes.c_uint64
            f.argtypes = [ctypes.c_uint64]
    return _lib


_OUT_SIZES = {
    "bls12_g1add": 128, "bls12_g1mul": 128, "bls12_g1multiexp": 128,
    "bls12_g1multiexp_naive": 128, "bls12_g1multiexp_bc": 128,
    "bls12_g2add": 256, "bls12_g2mul": 256, "bls12_g2multiexp": 256,
    "bls12_g2multiexp_naive": 256, "bls12_g2multiexp_bc": 256,
    "bls12_pairing": 32, "bls12_map_fp_to_g1": 128, "bls12_map_fp2_to_g2": 256,
}


def call(name, inp):
    inp = bytes(inp)
    out = ctypes.create_string_buffer(_OUT_SIZES[name])
    rc = getattr(lib(), "oracle_" + name)(out, inp, len(inp))
    return rc, (out.raw if rc == 0 else None)


def _le32(v):
    return int(v).to_bytes(32, "little")


def gen_msm_input(group, n, a, b, seed):
    """n records ([a+i*b]G || SplitMix64 scalar), EIP-encoded."""
    rec = 160 if group == "g1" else 288
    out = ctypes.create_string_buffer(n * rec)
    f = getattr(lib(), "oracle_gen_%s_msm_input" % group)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
    rc = f(out, n, _le32(a), _le32(b), seed)
    assert rc == 0
    return out.raw


def gen_pairing_input(k, a0, a1, b0, b1):
    out = ctypes.create_string_buffer(k * 384)
    f = lib().oracle_gen_pairing_input
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_char_p] * 4
    rc = f(out, k, _le32(a0), _le32(a1), _le32(b0), _le32(b1))
    assert rc == 0
    return out.raw


def in_subgroup(group, enc, slow=False):
    f = getattr(lib(), "oracle_%s_in_subgroup" % group)
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_char_p, ctypes.c_int]
    return f(bytes(enc), 1 if slow else 0)


def pairing_fp12(enc, final_exp=True):
    out = ctypes.create_string_buffer(576)
    f = lib().oracle_pairing_fp12
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    rc = f(out, bytes(enc), 1 if final_exp else 0)
    assert rc == 0, rc
    return out.raw


def g1_pippenger_mt(inp, threads):
    """All-cores bucket-method MSM: a context number for bench.py, NOT the reference's algorithm."""
    inp = bytes(inp)
    out = ctypes.create_string_buffer(128)
    f = lib().oracle_g1_pippenger_mt
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    rc = f(out, inp, len(inp), threads)
    return rc, (out.raw if rc == 0 else None)
