import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import oracle  # noqa: E402,F401  (adds oracle/pymodel to sys.path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def clib():
    from oracle import clib as c
    c.build()
    return c


@pytest.fixture(scope="session")
def X():
    import blst_eip2537_amd as pkg
    pkg.lib()
    return pkg.Eip2537Executor


def call_x(fn, inp):
    """(code, out|None) view of an executor method, like the C-ABI reports it."""
    from blst_eip2537_amd import Eip2537Error
    try:
        return 0, fn(inp)
    except Eip2537Error as e:
        return e.code, None
