"""RFC 9380 hash-to-curve for BLS12-381 G1 / G2 (SSWU + isogeny + cofactor clearing), big-integer
model.  TEST INFRASTRUCTURE ONLY (oracle).  This is what the reference obtains from
blst_map_to_g1(out, u, NULL) / blst_map_to_g2 (reference src/eip2537.c:1113,1155): map_to_curve
of ONE field element followed by cofactor clearing.  The isogeny tables are derived, not copied
(tools/derive_isogeny.py), and the whole construction is pinned by the RFC 9380 appendix J.9.1 /
J.10.1 test vectors in tests/test_h2c.py.
"""
import hashlib

from bls12_381 import *   # noqa: F401,F403
from bls12_381 import _try_fp

G1_A = 0x144698a3b8e9433d693a02c96d4982b0ea985383ee66a8d8e8981aefd881ac98936f8da0e0f97f5cf428082d584c1d
G1_B = 0x12e2908d11688030018b12e8753eee3b2016c1f0f24f4070a0b9c14fcef35ef55a23215a316ceaa5d1cc48e98e172be0
G1_Z = 11
G2_A = (0, 240)
G2_B = (1012, 1012)
G2_Z = ((-2) % P, (-1) % P)
H_EFF_G1 = 0xd201000000010001
H_EFF_G2 = H2 * (3 * Z * Z - 3)

ISO = {}          # filled by load_iso(): "g1"/"g2" -> (xnum, xden, ynum, yden)


def sgn0_fp(x):
    return x & 1


def sgn0_fp2(x):
    return (x[0] & 1) | ((x[0] == 0) & (x[1] & 1))


def _poly(F, coeffs, x):
    r = F.zero
    for c in reversed(coeffs):
        r = F.add(F.mul(r, x), c)
    return r


def sswu(F, A, B, Zc, u, sqrt, sgn0, is_zero):
    u2 = F.mul(u, u)
    zu2 = F.mul(Zc, u2)
    tv1 = F.add(F.mul(zu2, zu2), zu2)
    if is_zero(tv1):
        x1 = F.mul(B, F.inv(F.mul(Zc, A)))
    else:
        x1 = F.mul(F.mul(F.neg(B), F.inv(A)), F.add(F.one, F.inv(tv1)))
    gx1 = F.add(F.add(F.mul(F.mul(x1, x1), x1), F.mul(A, x1)), B)
    y = sqrt(gx1)
    if y is not None:
        x = x1
    else:
        x = F.mul(zu2, x1)
        gx2 = F.add(F.add(F.mul(F.mul(x, x), x), F.mul(A, x)), B)
        y = sqrt(gx2)
        assert y is not None
    if sgn0(u) != sgn0(y):
        y = F.neg(y)
    return (x, y)


def iso_map(F, which, pt):
    xnum, xden, ynum, yden = ISO[which]
    x, y = pt
    xd, yd = _poly(F, xden, x), _poly(F, yden, x)
    if xd == F.zero or yd == F.zero:
        return None                                   # kernel point -> infinity
    return (F.mul(_poly(F, xnum, x), F.inv(xd)), F.mul(y, F.mul(_poly(F, ynum, x), F.inv(yd))))


def map_to_curve_g1(u):
    return iso_map(FP, "g1", sswu(FP, G1_A, G1_B, G1_Z, u % P, fp_sqrt, sgn0_fp, lambda v: v == 0))


def map_to_curve_g2(u):
    return iso_map(FP2, "g2", sswu(FP2, G2_A, G2_B, G2_Z, u, f2_sqrt, sgn0_fp2, lambda v: v == F2_ZERO))


def map_fp_to_g1(u):
    """What bls12_map_fp_to_g1 computes: clear_cofactor(map_to_curve(u))."""
    return g1_mul(map_to_curve_g1(u), H_EFF_G1)


def map_fp2_to_g2(u):
    return g2_mul(map_to_curve_g2(u), H_EFF_G2)


def bls12_map_fp_to_g1(inp):
    if len(inp) != 64:
        raise EipError(INVALID_LENGTH)
    return encode_g1(map_fp_to_g1(decode_fp(inp)))


def bls12_map_fp2_to_g2(inp):
    if len(inp) != 128:
        raise EipError(INVALID_LENGTH)
    c0, c1 = _try_fp(inp[:64]), _try_fp(inp[64:])
    if c0 is None or c1 is None:
        raise EipError(INVALID_ELEMENT)
    return encode_g2(map_fp2_to_g2((c0, c1)))


# ---- hash_to_field / hash_to_curve (only used to check the RFC vectors)
def expand_message_xmd(msg, dst, n):
    h = hashlib.sha256
    ell = (n + 31) // 32
    dst_prime = dst + bytes([len(dst)])
    b0 = h(bytes(64) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    b = [h(b0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        b.append(h(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime).digest())
    return b"".join(b)[:n]


def hash_to_field(msg, dst, count, mdeg):
    L = 64
    data = expand_message_xmd(msg, dst, count * mdeg * L)
    out = []
    for i in range(count):
        e = [int.from_bytes(data[L * (j + i * mdeg):L * (j + i * mdeg) + L], "big") % P for j in range(mdeg)]
        out.append(e[0] if mdeg == 1 else tuple(e))
    return out


def hash_to_curve_g1(msg, dst):
    u = hash_to_field(msg, dst, 2, 1)
    return g1_mul(g1_add(map_to_curve_g1(u[0]), map_to_curve_g1(u[1])), H_EFF_G1), u


def hash_to_curve_g2(msg, dst):
    u = hash_to_field(msg, dst, 2, 2)
    return g2_mul(g2_add(map_to_curve_g2(u[0]), map_to_curve_g2(u[1])), H_EFF_G2), u


def load_iso():
    import iso_constants as c
    ISO["g1"] = (c.G1_XNUM, c.G1_XDEN, c.G1_YNUM, c.G1_YDEN)
    ISO["g2"] = (c.G2_XNUM, c.G2_XDEN, c.G2_YNUM, c.G2_YDEN)


try:
    load_iso()
except ImportError:      # before tools/derive_isogeny.py has run
    pass
