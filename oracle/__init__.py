"""CPU oracle for the EIP-2537 hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/README.md).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product package
blst_eip2537_amd never does.

  oracle.clib      ctypes binding of oracle/liboracle_eip2537.so (C restatement of
                   the reference's src/eip2537.c control flow)
  oracle.pymodel   big-integer Python model (slow; golden generation, small cases)
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PM = os.path.join(_HERE, "pymodel")
if _PM not in sys.path:
    sys.path.insert(0, _PM)
