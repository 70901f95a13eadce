"""The constructible EIP-2537 vector families (tests/golden/eip2537_constructed/, written by
tools/gen_eip_vectors.py from the big-integer Python model in the reference's own CSV / JSON formats and
file names) through tools/run_kat.py -- the ingestion path the published files would take: success rows
byte-for-byte, failure files by the error class the reference's C harness requires
(src/test.c:144-165, 337-358, 481-511, 564-585, 638-659), multiexp rows through _bc / _naive as well
(src/test.c:208-228).  CPU: the oracle on everything and the product's host precompiles; GPU: the
product's multiexp / pairing entry points."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_kat  # noqa: E402

DIR = os.path.join(ROOT, "tests", "golden", "eip2537_constructed")
HOST_OPS = {"g1_add", "g1_mul", "g2_add", "g2_mul", "map_fp_to_g1", "map_fp2_to_g2"}
GPU_OPS = {"g1_multiexp", "g2_multiexp", "pairing"}
FILES = sorted(os.listdir(DIR))


class _OracleExecutor:
    """run_kat's executor interface over the CPU oracle (all 13 precompiles run on the CPU there)."""
    def __init__(self, clib):
        self.clib = clib

    def __getattr__(self, op):
        from blst_eip2537_amd import Eip2537Error
        name = "bls12_" + op.replace("g1_", "g1").replace("g2_", "g2")

        def call(inp):
            rc, out = self.clib.call(name, inp)
            if rc:
                raise Eip2537Error(rc)
            return out
        return call


def test_fixture_set_has_the_reference_file_names():
    want = {"g1_add.csv", "g1_mul.csv", "g1_multiexp.csv", "g1_not_on_curve.csv", "g2_add.csv", "g2_mul.csv", "g2_multiexp.csv",
            "g2_not_on_curve.csv", "fp_to_g1.csv", "fp2_to_g2.csv", "pairing.csv", "invalid_subgroup_for_pairing.csv",
            "invalid_fp_encoding.csv", "invalid_fp2_encoding.csv"}                      # reference build.sh:17-30
    for stem in ["G1Add", "G1Mul", "G1MultiExp", "G2Add", "G2Mul", "G2MultiExp", "MapG1", "MapG2", "Pairing"]:
        want |= {"bls%s.json" % stem, "fail-bls%s.json" % stem}                         # reference build.sh:32-49
    assert set(FILES) == want


@pytest.mark.parametrize("name", FILES)
def test_oracle_on_constructed_vectors(clib, name):
    cases = run_kat.load(os.path.join(DIR, name))
    ok, bad = run_kat.run(cases, _OracleExecutor(clib))
    assert not bad and ok >= len(cases), (name, bad[:3])


@pytest.mark.parametrize("name", FILES)
def test_product_host_precompiles_on_constructed_vectors(X, name):
    cases = run_kat.load(os.path.join(DIR, name))
    ok, bad = run_kat.run(cases, X, ops=HOST_OPS)
    assert not bad, (name, bad[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("route", [-1, 0])
@pytest.mark.parametrize("name", [f for f in FILES if any(k in f.lower() for k in ("multiexp", "pairing"))])
def test_product_gpu_precompiles_on_constructed_vectors(X, name, route):
    cases = run_kat.load(os.path.join(DIR, name))
    X.set_route(route)
    try:
        ok, bad = run_kat.run(cases, X, ops=GPU_OPS)
    finally:
        X.set_route(-1)
    assert not bad and ok >= len(cases), (name, bad[:3])
