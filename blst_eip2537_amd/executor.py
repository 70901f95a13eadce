"""ctypes host layer over libeip2537_hip.so.

Mirrors the reference bindings one-to-one (reference rust/src/lib.rs:98-328,
go/blst_eip2537.go:44-211): every method takes the EIP-2537 encoded input bytes and returns the
encoded output bytes, or raises Eip2537Error whose message is the reference's error string
(rust/src/lib.rs:101-113).  The device-resident and sharded entry points of
include/eip2537_hip.h are exposed as *_dev / *_partial_dev / *_combine.
"""
import ctypes
import os

# Concurrent callers drive up to 8 engine slots x 3 streams; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (4 unless told otherwise) when it initialises, so the variable has to be in the environment before the first HIP
# call of the process.  The library itself no longer touches the environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
# EIP2537_HIP_LIB: load another build of the same library (kernel A/B runs); default is the in-tree one
_SO = os.environ.get("EIP2537_HIP_LIB") or os.path.join(_HERE, "libeip2537_hip.so")

ERROR_STRINGS = {
    0: "Success",
    1: "point not on curve",
    2: "point not in subgroup",
    3: "invalid element",
    4: "encoding error",
    5: "invalid length",
    6: "empty input",
    7: "memory allocation error",
}


class Eip2537Error(Exception):
    def __init__(self, code):
        self.code = int(code)
        super().__init__(ERROR_STRINGS.get(self.code, "unknown error condition"))


_ABI = {  # name -> output bytes          (reference src/eip2537.h:42-59)
    "bls12_g1add": 128, "bls12_g1mul": 128,
    "bls12_g1multiexp": 128, "bls12_g1multiexp_naive": 128, "bls12_g1multiexp_bc": 128,
    "bls12_g2add": 256, "bls12_g2mul": 256,
    "bls12_g2multiexp": 256, "bls12_g2multiexp_naive": 256, "bls12_g2multiexp_bc": 256,
    "bls12_pairing": 32, "bls12_map_fp_to_g1": 128, "bls12_map_fp2_to_g2": 256,
}
_GAS_FIXED = ["bls12_g1add_gas", "bls12_g1mul_gas", "bls12_g2add_gas", "bls12_g2mul_gas",
              "bls12_map_fp_to_g1_gas", "bls12_map_fp2_to_g2_gas"]
_GAS_LEN = ["bls12_g1multiexp_gas", "bls12_g2multiexp_gas", "bls12_pairing_gas"]
_DEV = {  # name -> output bytes          (include/eip2537_hip.h)
    "eip2537_hip_g1multiexp_dev": 128, "eip2537_hip_g2multiexp_dev": 256, "eip2537_hip_pairing_dev": 32,
    "eip2537_hip_g1msm_partial_dev": 192, "eip2537_hip_g2msm_partial_dev": 384,
    "eip2537_hip_pairing_partial_dev": 576,
}
_COMBINE = {  # name -> (partial bytes, output bytes)
    "eip2537_hip_g1msm_combine": (192, 128), "eip2537_hip_g2msm_combine": (384, 256),
    "eip2537_hip_pairing_combine": (576, 32),
}

_lib = None
_bound_runtime = "system"


def lib_path():
    return _SO


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (torch/lib); the engine
    library asks for the same soname and would otherwise map /opt/rocm's copy when it is loaded first -- and whichever of the
    two runtimes opens the device second then reports "no ROCm-capable device".  So when PyTorch is installed but not imported
    yet, its copy is mapped first (without importing torch) and the loader binds the engine to it, exactly as it does when
    `import torch` came first.  Without PyTorch the engine uses the ROCm installation's runtime.
    EIP2537_HIP_PRELOAD_TORCH_RUNTIME=0 turns the preload off (a process that will never import torch and wants the ROCm
    installation's runtime); EIP2537_HIP_VERBOSE=1 logs which copy was bound."""
    import importlib.util
    import sys
    global _bound_runtime
    if os.environ.get("EIP2537_HIP_PRELOAD_TORCH_RUNTIME", "1") == "0":
        _bound_runtime = "system (preload disabled)"
        return
    if "torch" in sys.modules:
        _bound_runtime = "torch (imported before the engine)"
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            _bound_runtime = cand
        except OSError as ex:
            _bound_runtime = "system (mapping %s failed: %s)" % (cand, ex)
            sys.stderr.write("[eip2537_hip] note: could not map PyTorch's HIP runtime %s (%s); the engine binds the system copy\n" % (cand, ex))
    if os.environ.get("EIP2537_HIP_VERBOSE") == "1":
        sys.stderr.write("[eip2537_hip] HIP runtime bound for the engine: %s\n" % _bound_runtime)


def bound_hip_runtime():
    """Which libamdhip64 the engine library was bound to when it was loaded (a path, or a description)."""
    lib()
    return _bound_runtime


def lib():
    """Load the in-tree engine library; fail loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise RuntimeError(
            "libeip2537_hip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C blst_eip2537_amd/csrc`.  There is no CPU fallback." % _SO)
    _one_hip_runtime()
    L = ctypes.CDLL(_SO)
    for name in _ABI:
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    for name in _GAS_FIXED:
        getattr(L, name).restype = ctypes.c_uint64
        getattr(L, name).argtypes = []
    for name in _GAS_LEN:
        getattr(L, name).restype = ctypes.c_uint64
        getattr(L, name).argtypes = [ctypes.c_uint64]
    for name in _DEV:
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    for name in _COMBINE:
        f = getattr(L, name)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.eip2537_hip_init.restype = ctypes.c_int
    L.eip2537_hip_init.argtypes = [ctypes.c_int]
    L.eip2537_hip_coalesce_stats.restype = None
    L.eip2537_hip_coalesce_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 3
    L.eip2537_hip_set_route.restype = ctypes.c_int
    L.eip2537_hip_set_route.argtypes = [ctypes.c_int]
    L.eip2537_hip_set_window.restype = ctypes.c_int
    L.eip2537_hip_set_window.argtypes = [ctypes.c_int]
    L.eip2537_hip_field_selftest.restype = ctypes.c_int
    L.eip2537_hip_field_selftest.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    L.eip2537_hip_limb_selftest.restype = ctypes.c_int
    L.eip2537_hip_limb_selftest.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    L.eip2537_hip_device_count.restype = ctypes.c_int
    L.eip2537_hip_device_count.argtypes = []
    L.eip2537_hip_trim.restype = ctypes.c_size_t
    L.eip2537_hip_trim.argtypes = [ctypes.c_size_t]
    L.eip2537_hip_last_plan.restype = ctypes.c_int
    L.eip2537_hip_last_plan.argtypes = [ctypes.c_char_p, ctypes.c_size_t] + [ctypes.POINTER(ctypes.c_int)] * 3 + \
                                       [ctypes.POINTER(ctypes.c_uint32)] * 2
    L.eip2537_hip_last_shards.restype = ctypes.c_int
    L.eip2537_hip_last_shards.argtypes = []
    L.eip2537_hip_last_timing_aux.restype = None
    L.eip2537_hip_last_timing_aux.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.eip2537_hip_last_timing.restype = None
    L.eip2537_hip_last_timing.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    _lib = L
    return L


def _call(name, inp):
    if not isinstance(inp, bytes):
        inp = bytes(inp)
    out = ctypes.create_string_buffer(_ABI[name])
    # the bytes object is passed as is (no copy: the library never writes `in`); a zero-length
    # input becomes a non-null dangling pointer like Rust's (rust/src/lib.rs:147), which the
    # library must not dereference.
    buf = ctypes.cast(ctypes.c_char_p(inp), ctypes.c_void_p) if inp else ctypes.create_string_buffer(1)
    rc = getattr(lib(), name)(out, buf, len(inp))
    if rc != 0:
        raise Eip2537Error(rc)
    return out.raw


class Eip2537Executor:
    """Same surface as the reference's ``blstEIP2537Executor`` (rust/src/lib.rs:98)."""

    @staticmethod
    def g1_add(inp): return _call("bls12_g1add", inp)
    @staticmethod
    def g1_mul(inp): return _call("bls12_g1mul", inp)
    @staticmethod
    def g1_multiexp(inp): return _call("bls12_g1multiexp", inp)
    @staticmethod
    def g1_multiexp_naive(inp): return _call("bls12_g1multiexp_naive", inp)
    @staticmethod
    def g1_multiexp_bc(inp): return _call("bls12_g1multiexp_bc", inp)
    @staticmethod
    def g2_add(inp): return _call("bls12_g2add", inp)
    @staticmethod
    def g2_mul(inp): return _call("bls12_g2mul", inp)
    @staticmethod
    def g2_multiexp(inp): return _call("bls12_g2multiexp", inp)
    @staticmethod
    def g2_multiexp_naive(inp): return _call("bls12_g2multiexp_naive", inp)
    @staticmethod
    def g2_multiexp_bc(inp): return _call("bls12_g2multiexp_bc", inp)
    @staticmethod
    def pairing(inp): return _call("bls12_pairing", inp)
    @staticmethod
    def map_fp_to_g1(inp): return _call("bls12_map_fp_to_g1", inp)
    @staticmethod
    def map_fp2_to_g2(inp): return _call("bls12_map_fp2_to_g2", inp)

    # ---- gas schedule (reference src/eip2537.c:1168-1271)
    @staticmethod
    def gas(name, input_len=None):
        f = getattr(lib(), "bls12_%s_gas" % name)
        return int(f() if input_len is None else f(input_len))

    # ---- device-resident / sharded extensions (include/eip2537_hip.h)
    @staticmethod
    def init(device=-1):
        rc = lib().eip2537_hip_init(device)
        if rc != 0:
            raise Eip2537Error(rc)

    @staticmethod
    def dev_call(name, dev_ptr, n_records):
        """name in _DEV; dev_ptr = integer device address of the encoded records."""
        out = ctypes.create_string_buffer(_DEV[name])
        rc = getattr(lib(), name)(out, ctypes.c_void_p(dev_ptr), n_records)
        if rc != 0:
            raise Eip2537Error(rc)
        return out.raw

    @staticmethod
    def combine(name, partials):
        psz, osz = _COMBINE[name]
        blob = b"".join(partials)
        assert len(blob) % psz == 0
        out = ctypes.create_string_buffer(osz)
        rc = getattr(lib(), name)(out, blob, len(blob) // psz)
        if rc != 0:
            raise Eip2537Error(rc)
        return out.raw

    @staticmethod
    def gen_msm_input(group, n, a, b, seed, start=0):
        """Synthetic MSM records, generated by the library's own host code."""
        rec = 160 if group == "g1" else 288
        out = ctypes.create_string_buffer(n * rec)
        f = getattr(lib(), "eip2537_hip_gen_%s_msm_input" % group)
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64]
        rc = f(out, n, int(a).to_bytes(32, "little"), int(b).to_bytes(32, "little"), seed, start)
        assert rc == 0
        return out.raw

    @staticmethod
    def gen_pairing_input(k, a0, a1, b0, b1, start=0):
        out = ctypes.create_string_buffer(k * 384)
        f = lib().eip2537_hip_gen_pairing_input
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_char_p] * 4 + [ctypes.c_uint64]
        rc = f(out, k, *[int(v).to_bytes(32, "little") for v in (a0, a1, b0, b1)], start)
        assert rc == 0
        return out.raw

    @staticmethod
    def last_timing():
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        lib().eip2537_hip_last_timing(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    @staticmethod
    def last_timing_aux():
        """Two more device intervals of the last GPU call in ms (eip2537_hip.h): pairing (G1 membership kernel, line
        products), MSM (sort stage, fold + bucket reduce)."""
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        lib().eip2537_hip_last_timing_aux(ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    @staticmethod
    def last_plan():
        """What the last GPU call ran: dict(kernel, window_bits, windows, lanes, units, buckets) or None."""
        name = ctypes.create_string_buffer(64)
        c, w, ln = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        u, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
        if lib().eip2537_hip_last_plan(name, 64, ctypes.byref(c), ctypes.byref(w), ctypes.byref(ln),
                                       ctypes.byref(u), ctypes.byref(b)) != 0:
            return None
        return {"kernel": name.value.decode(), "window_bits": c.value, "windows": w.value, "lanes": ln.value,
                "units": u.value, "buckets": b.value, "shards": int(lib().eip2537_hip_last_shards())}

    @staticmethod
    def device_count():
        return int(lib().eip2537_hip_device_count())

    @staticmethod
    def trim(keep_bytes=0):
        return int(lib().eip2537_hip_trim(keep_bytes))

    @staticmethod
    def coalesce_stats():
        """(device pipelines run for small multiexp calls, calls served, largest batch) since load."""
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        lib().eip2537_hip_coalesce_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    @staticmethod
    def set_route(route):
        """-1: default crossover; 0: always the GPU; 1: the library's host code up to 64 units."""
        rc = lib().eip2537_hip_set_route(route)
        if rc != 0:
            raise Eip2537Error(rc)

    @staticmethod
    def field_selftest(seed, n):
        """Mismatch counts (product, square, lazy product, lazy square) of the device Fp products on n
        operand pairs against the independent 12 x 32-bit product (eip2537_hip.h)."""
        bad = (ctypes.c_uint64 * 4)()
        rc = lib().eip2537_hip_field_selftest(seed, n, bad)
        if rc != 0:
            raise Eip2537Error(rc)
        return tuple(bad)

    @staticmethod
    def limb_selftest(seed, n):
        """Mismatch counts of the limb-form device primitives (eip2537_hip.h: mulL, sqrL, mul2L, fp_mul2_cols30, limb
        conversion, weak reduction / shifts, sums and differences, zero test) against the 12 x 32-bit product."""
        bad = (ctypes.c_uint64 * 8)()
        rc = lib().eip2537_hip_limb_selftest(seed, n, bad)
        if rc != 0:
            raise Eip2537Error(rc)
        return tuple(bad)

    @staticmethod
    def set_window(c):
        rc = lib().eip2537_hip_set_window(c)
        if rc != 0:
            raise Eip2537Error(rc)
