"""blst_eip2537_amd -- MI355X-native EIP-2537 precompile engine.

The product is the C-ABI shared library ``libeip2537_hip.so`` (HIP/gfx950 kernels + host code,
built from ``csrc/``; headers in ``include/``).  This package is the thin Python host layer over
that C-ABI, mirroring the reference's own bindings (rust/src/lib.rs ``blstEIP2537Executor``,
go/blst_eip2537.go ``G1Add`` ... ``MapFp2ToG2``): same operation names, bytes in, bytes out,
an exception carrying the reference's error string on failure.

There is no CPU fallback for the multiexp / pairing path: if the library is missing, importing
``executor`` raises; if no HIP device is present those calls return ``memory allocation error``.
"""
from .executor import (  # noqa: F401
    Eip2537Error, Eip2537Executor, ERROR_STRINGS, lib, lib_path, bound_hip_runtime,
)
