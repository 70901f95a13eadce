"""CPU tests of the oracle (no GPU): the C restatement against the golden fixtures produced by
the independent Python big-integer model, and the structural identities that pin the model."""
import json
import os

import pytest

import bls12_381 as m

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
A = 0x1f3a5c7e9b2d4f6081a3c5e7092b4d6f8ea1c3e5a7092b4d6f80a2c4e6
B = 0x0123456789abcdef0fedcba987654321


def _kat():
    with open(os.path.join(GOLD, "kat.json")) as f:
        return json.load(f)


def test_oracle_matches_kat(clib):
    n = 0
    for v in _kat():
        inp = bytes.fromhex(v["input"])
        want = (v["code"], bytes.fromhex(v["output"]) if v["output"] is not None else None)
        names = ["bls12_" + v["op"]]
        if v["op"].endswith("multiexp"):
            names += [names[0] + "_naive", names[0] + "_bc"]     # reference tests all three (src/test.c:208-228)
        for name in names:
            got = clib.call(name, inp)
            if name.endswith("_bc") and v["code"] != 0 and len(inp) and len(inp) % (160 if "g1" in name else 288) == 0:
                assert got[0] == v["code"], (name, v["op"])
            else:
                assert got == want, (name, v["op"], n)
        n += 1
    assert n >= 60


@pytest.mark.parametrize("wl,log2n", [("g1msm", 10), ("g1msm", 16), ("g2msm", 10)])
def test_oracle_analytic_golden(clib, wl, log2n):
    group = "g1" if wl == "g1msm" else "g2"
    seed = 0x25370000 + (0 if wl == "g1msm" else 0x100) + log2n
    inp = clib.gen_msm_input(group, 1 << log2n, A, B, seed)
    with open(os.path.join(GOLD, "%s_2p%d.hex" % (wl, log2n))) as f:
        want = bytes.fromhex(f.read().strip())
    assert clib.call("bls12_%smultiexp" % group, inp) == (0, want)


def test_model_constants_and_identities():
    z = m.Z
    assert (z - 1) ** 2 * (z ** 4 - z ** 2 + 1) // 3 + z == m.P
    assert z ** 4 - z ** 2 + 1 == m.R
    assert m.ec_on_curve(m.FP, m.B1, m.G1) and m.ec_on_curve(m.FP2, m.B2, m.G2)
    assert m.g1_mul(m.G1, m.R) is None and m.g2_mul(m.G2, m.R) is None
    assert m.g1_mul((0, 2), 3) is None and m.g1_mul((0, 2), m.R) is not None      # order-3, outside G1


def test_model_pairing_bilinear_and_fast_forms():
    import fastmodel as fm
    P, Q = m.g1_mul(m.G1, 1234567), m.g2_mul(m.G2, 7654321)
    e = m.pairing(m.G1, m.G2)
    assert e != m.F12_ONE and m.f12_pow(e, m.R) == m.F12_ONE
    f = fm.miller_loop_fast(P, Q)
    assert m.final_exp(f) == m.f12_pow(e, 1234567 * 7654321 % m.R)
    assert fm.final_exp_fast(f) == m.f12_pow(m.final_exp(f), 3)


def test_oracle_pairing_value_matches_model(clib):
    P, Q = m.g1_mul(m.G1, 99991), m.g2_mul(m.G2, 31337)
    out = clib.pairing_fp12(m.encode_g1(P) + m.encode_g2(Q), True)
    e3 = m.f12_pow(m.pairing(P, Q), 3)
    flat = [c for c6 in e3 for c2 in c6 for c in c2]
    assert out == b"".join(int(c).to_bytes(48, "big") for c in flat)


def _torsion(gen, mul, n_order, l, e, rng):
    while True:
        t = mul(gen(rng, False), n_order // l ** e)
        if t is not None:
            return t


def test_subgroup_checks_sound_on_cofactor_torsion(clib):
    """The endomorphism tests must reject every prime-power torsion component of both cofactors."""
    rng = m.SplitMix64(77)
    P1, Q1 = m.g1_mul(m.G1, 5), m.g2_mul(m.G2, 5)
    for l, e in [(3, 1), (11, 2), (10177, 2), (859267, 2), (52437899, 2)]:
        assert m.H1 % l ** e == 0
        t = _torsion(m.random_g1, m.g1_mul, m.H1 * m.R, l, e, rng)
        for pt in (t, m.g1_add(t, P1)):
            assert clib.in_subgroup("g1", m.encode_g1(pt), False) == 0
            assert clib.in_subgroup("g1", m.encode_g1(pt), True) == 0
    for l, e in [(13, 2), (23, 2), (2713, 1), (11953, 1), (262069, 1)]:
        assert m.H2 % l ** e == 0
        t = _torsion(m.random_g2, m.g2_mul, m.H2 * m.R, l, e, rng)
        for pt in (t, m.g2_add(t, Q1)):
            assert clib.in_subgroup("g2", m.encode_g2(pt), False) == 0
            assert clib.in_subgroup("g2", m.encode_g2(pt), True) == 0
    assert clib.in_subgroup("g1", m.encode_g1(P1), False) == 1
    assert clib.in_subgroup("g2", m.encode_g2(Q1), False) == 1


def test_oracle_gas(clib):
    L = clib.lib()
    assert L.oracle_g1multiexp_gas(160) == 12000 * 1200 // 1000
    assert L.oracle_g1multiexp_gas(160 * 128) == 128 * 12000 * 174 // 1000
    assert L.oracle_g1multiexp_gas(160 * 4096) == 4096 * 12000 * 174 // 1000
    assert L.oracle_g2multiexp_gas(288 * 2) == 2 * 55000 * 888 // 1000
    assert L.oracle_pairing_gas(384 * 3) == 115000 + 3 * 23000
    assert L.oracle_pairing_gas(100) == 0
