"""Big-integer model of the EIP-2537 precompiles over BLS12-381.

TEST INFRASTRUCTURE ONLY.  This module is the slow, "obviously correct" second
implementation used to (a) cross-check the C oracle in oracle/c/ and (b) generate
the golden fixtures under tests/golden/.  Nothing in the shipped product path
imports it.

PARITY UNPINNED: the reference (sean-sn/blst_eip2537) delegates all arithmetic to
supranational/blst, which is not vendored under /root/reference and cannot be
built offline, and the reference's own known-answer vectors are downloaded at
build time (build.sh:13-52) and are absent.  Parity is therefore argued from the
canonical encoding of every output ("parity principle") plus
agreement of three independent implementations (this file, oracle/c, the HIP
engine) and algebraic identities.

Everything here follows the wire format and validation order of the reference:
  fp_from_bytes        src/eip2537.c:263-309
  decode_g1_point      src/eip2537.c:320-343
  decode_g2_point      src/eip2537.c:381-404
  decode_scalar        src/eip2537.c:417-420
  bls12_g1add/mul      src/eip2537.c:434-524
  bls12_g1multiexp     src/eip2537.c:541-616
  bls12_g2*            src/eip2537.c:722-905
  bls12_pairing        src/eip2537.c:1020-1081
The arithmetic itself is textbook: affine short-Weierstrass group law with modular
inversion, Fp2/Fp6/Fp12 tower, untwist-then-Miller-loop Tate/ate pairing with a
plain exponentiation by (p^12-1)/r.
"""

# ----------------------------------------------------------------------------
# Parameters
# ----------------------------------------------------------------------------
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
Z_ABS = 0xd201000000010000          # the BLS parameter is z = -Z_ABS
Z = -Z_ABS
H1 = 0x396c8c005555e1568c00aaab0000aaab   # G1 cofactor (z-1)^2/3
# G2 cofactor
H2 = 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5

G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2_X = (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
        0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e)
G2_Y = (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
        0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)

# error codes, src/eip2537.h:31-40
SUCCESS, NOT_ON_CURVE, NOT_IN_SUBGROUP, INVALID_ELEMENT, ENCODING_ERROR, \
    INVALID_LENGTH, EMPTY_INPUT, MEMORY_ERROR = range(8)


class EipError(Exception):
    def __init__(self, code):
        super().__init__("EIP2537 error %d" % code)
        self.code = code


# ----------------------------------------------------------------------------
# Fp2 = Fp[u]/(u^2+1)      elements are (c0, c1)
# ----------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)
def f2_muls(a, s): return (a[0] * s % P, a[1] * s % P)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)                           # xi = 1 + u, the sextic non-residue


def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


def fp_sqrt(a):
    """sqrt in Fp (p = 3 mod 4) or None."""
    a %= P
    s = pow(a, (P + 1) // 4, P)
    return s if s * s % P == a else None


def f2_sqrt(a):
    """sqrt in Fp2 or None (complex method: norm, then half-trace)."""
    if a == F2_ZERO:
        return F2_ZERO
    a0, a1 = a
    if a1 == 0:
        s = fp_sqrt(a0)
        if s is not None:
            return (s, 0)
        s = fp_sqrt((-a0) % P)
        return (0, s)
    n = fp_sqrt((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    inv2 = pow(2, -1, P)
    for nn in (n, (-n) % P):
        t = (a0 + nn) * inv2 % P
        x0 = fp_sqrt(t)
        if x0 is None or x0 == 0:
            continue
        x1 = a1 * pow(2 * x0, -1, P) % P
        if f2_sqr((x0, x1)) == a:
            return (x0, x1)
    return None


# ----------------------------------------------------------------------------
# Fp6 = Fp2[v]/(v^3 - xi)   elements are (a0, a1, a2)
# Fp12 = Fp6[w]/(w^2 - v)   elements are (c0, c1)
# ----------------------------------------------------------------------------
def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = f2_mul(a0, b0)
    t1 = f2_add(f2_mul(a0, b1), f2_mul(a1, b0))
    t2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    t3 = f2_add(f2_mul(a1, b2), f2_mul(a2, b1))
    t4 = f2_mul(a2, b2)
    return (f2_add(t0, f2_mul(t3, XI)), f2_add(t1, f2_mul(t4, XI)), t2)
def f6_mul_by_v(a): return (f2_mul(a[2], XI), a[0], a[1])
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_inv(a):
    a0, a1, a2 = a
    c0 = f2_sub(f2_sqr(a0), f2_mul(XI, f2_mul(a1, a2)))
    c1 = f2_sub(f2_mul(XI, f2_sqr(a2)), f2_mul(a0, a1))
    c2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    t = f2_add(f2_mul(a0, c0), f2_mul(XI, f2_add(f2_mul(a2, c1), f2_mul(a1, c2))))
    ti = f2_inv(t)
    return (f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti))


def f12_add(a, b): return (f6_add(a[0], b[0]), f6_add(a[1], b[1]))
def f12_sub(a, b): return (f6_sub(a[0], b[0]), f6_sub(a[1], b[1]))
def f12_neg(a): return (f6_neg(a[0]), f6_neg(a[1]))
def f12_mul(a, b):
    t0 = f6_mul(a[0], b[0])
    t1 = f6_mul(a[1], b[1])
    c0 = f6_add(t0, f6_mul_by_v(t1))
    c1 = f6_add(f6_mul(a[0], b[1]), f6_mul(a[1], b[0]))
    return (c0, c1)
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    t = f6_sub(f6_mul(a[0], a[0]), f6_mul_by_v(f6_mul(a[1], a[1])))
    ti = f6_inv(t)
    return (f6_mul(a[0], ti), f6_neg(f6_mul(a[1], ti)))
F12_ZERO = (F6_ZERO, F6_ZERO)
F12_ONE = (F6_ONE, F6_ZERO)


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_sqr(a)
        e >>= 1
    return r


def f12_from_f2(a): return ((a, F2_ZERO, F2_ZERO), F6_ZERO)
def f12_from_fp(a): return f12_from_f2((a % P, 0))
F12_W = (F6_ZERO, F6_ONE)                                   # w
F12_W2 = ((F2_ZERO, F2_ONE, F2_ZERO), F6_ZERO)             # w^2 = v
F12_W3 = (F6_ZERO, (F2_ZERO, F2_ONE, F2_ZERO))             # w^3 = v*w


# ----------------------------------------------------------------------------
# Generic affine short-Weierstrass arithmetic, y^2 = x^3 + b, infinity = None
# ----------------------------------------------------------------------------
class _Field:
    def __init__(s, add, sub, mul, inv, neg, zero, one):
        s.add, s.sub, s.mul, s.inv, s.neg, s.zero, s.one = add, sub, mul, inv, neg, zero, one


FP = _Field(lambda a, b: (a + b) % P, lambda a, b: (a - b) % P, lambda a, b: a * b % P,
            lambda a: pow(a, -1, P), lambda a: (-a) % P, 0, 1)
FP2 = _Field(f2_add, f2_sub, f2_mul, f2_inv, f2_neg, F2_ZERO, F2_ONE)
FP12 = _Field(f12_add, f12_sub, f12_mul, f12_inv, f12_neg, F12_ZERO, F12_ONE)
B1 = 4
B2 = (4, 4)


def ec_on_curve(F, b, pt):
    if pt is None:
        return True
    x, y = pt
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), b)


def ec_neg(F, pt):
    return None if pt is None else (pt[0], F.neg(pt[1]))


def ec_add(F, p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        xx = F.mul(x1, x1)
        lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_mul(F, pt, k):
    """True scalar multiplication [k]P on the full curve, any non-negative k."""
    acc = None
    while k:
        if k & 1:
            acc = ec_add(F, acc, pt)
        pt = ec_add(F, pt, pt)
        k >>= 1
    return acc


G1 = (G1_X, G1_Y)
G2 = (G2_X, G2_Y)


def g1_add(a, b): return ec_add(FP, a, b)
def g1_mul(a, k): return ec_mul(FP, a, k)
def g2_add(a, b): return ec_add(FP2, a, b)
def g2_mul(a, k): return ec_mul(FP2, a, k)
def g1_in_subgroup(a): return ec_mul(FP, a, R) is None
def g2_in_subgroup(a): return ec_mul(FP2, a, R) is None


# ----------------------------------------------------------------------------
# Pairing (slow, generic): untwist Q to E(Fp12), affine Miller loop, pow.
# ----------------------------------------------------------------------------
_W2_INV = f12_inv(F12_W2)
_W3_INV = f12_inv(F12_W3)


def untwist(q):
    """E'(Fp2) -> E(Fp12):  (x', y') -> (x'/w^2, y'/w^3)   (M-type twist)."""
    x, y = q
    return (f12_mul(f12_from_f2(x), _W2_INV), f12_mul(f12_from_f2(y), _W3_INV))


def _line(F, t, q, at):
    """Line through t and q (tangent if equal) evaluated at `at`; all in E(Fp12)."""
    x1, y1 = t
    x2, y2 = q
    xp, yp = at
    if x1 != x2:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    elif y1 == y2:
        xx = F.mul(x1, x1)
        lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(y1, y1)))
    else:
        return F.sub(xp, x1)
    return F.sub(F.sub(yp, y1), F.mul(lam, F.sub(xp, x1)))


def miller_loop(p, q):
    """f_{|z|,Q}(P), conjugated because z < 0.  p in E(Fp), q in E'(Fp2)."""
    if p is None or q is None:
        return F12_ONE
    P12 = (f12_from_fp(p[0]), f12_from_fp(p[1]))
    Q12 = untwist(q)
    F = FP12
    f = F12_ONE
    t = Q12
    for i in range(Z_ABS.bit_length() - 2, -1, -1):
        f = f12_mul(f12_sqr(f), _line(F, t, t, P12))
        t = ec_add(F, t, t)
        if (Z_ABS >> i) & 1:
            f = f12_mul(f, _line(F, t, Q12, P12))
            t = ec_add(F, t, Q12)
    return f12_conj(f)


FINAL_EXP = (P ** 12 - 1) // R


def final_exp(f):
    return f12_pow(f, FINAL_EXP)


def pairing(p, q):
    return final_exp(miller_loop(p, q))


# ----------------------------------------------------------------------------
# EIP-2537 wire codec (reference: src/eip2537.c:263-420)
# ----------------------------------------------------------------------------
def decode_fp(b):
    """64 bytes -> int; raises INVALID_ELEMENT like fp_from_bytes returning -1."""
    assert len(b) == 64
    if any(b[:16]):
        raise EipError(INVALID_ELEMENT)
    v = int.from_bytes(b[16:], "big")
    if v >= P:
        raise EipError(INVALID_ELEMENT)
    return v


def encode_fp(v):
    return bytes(16) + int(v).to_bytes(48, "big")


def _try_fp(b):
    try:
        return decode_fp(b)
    except EipError:
        return None


def decode_g1(b):
    assert len(b) == 128
    x, y = _try_fp(b[:64]), _try_fp(b[64:])
    if x is None or y is None:          # both decoded before the verdict (:322-328)
        raise EipError(INVALID_ELEMENT)
    if x == 0 and y == 0:
        return None
    if not ec_on_curve(FP, B1, (x, y)):
        raise EipError(NOT_ON_CURVE)
    return (x, y)


def encode_g1(pt):
    if pt is None:
        return bytes(128)
    return encode_fp(pt[0]) + encode_fp(pt[1])


def decode_g2(b):
    assert len(b) == 256
    c = [_try_fp(b[i * 64:(i + 1) * 64]) for i in range(4)]
    if any(v is None for v in c):
        raise EipError(INVALID_ELEMENT)
    x, y = (c[0], c[1]), (c[2], c[3])
    if x == F2_ZERO and y == F2_ZERO:
        return None
    if not ec_on_curve(FP2, B2, (x, y)):
        raise EipError(NOT_ON_CURVE)
    return (x, y)


def encode_g2(pt):
    if pt is None:
        return bytes(256)
    (x0, x1), (y0, y1) = pt
    return encode_fp(x0) + encode_fp(x1) + encode_fp(y0) + encode_fp(y1)


def decode_scalar(b):
    assert len(b) == 32
    return int.from_bytes(b, "big")     # never fails, not reduced (:417-420)


def encode_scalar(k):
    return int(k).to_bytes(32, "big")


# ----------------------------------------------------------------------------
# Precompiles: bytes -> bytes, raising EipError(code)
# ----------------------------------------------------------------------------
def bls12_g1add(inp):
    if len(inp) != 256:
        raise EipError(INVALID_LENGTH)
    a = decode_g1(inp[:128])
    b = decode_g1(inp[128:])
    return encode_g1(g1_add(a, b))


def bls12_g1mul(inp):
    if len(inp) != 160:
        raise EipError(INVALID_LENGTH)
    a = decode_g1(inp[:128])
    k = decode_scalar(inp[128:])
    return encode_g1(g1_mul(a, k))


def bls12_g1multiexp(inp):
    if len(inp) == 0 or len(inp) % 160:
        raise EipError(INVALID_LENGTH)
    acc = None
    for o in range(0, len(inp), 160):
        a = decode_g1(inp[o:o + 128])
        k = decode_scalar(inp[o + 128:o + 160])
        acc = g1_add(acc, g1_mul(a, k))
    return encode_g1(acc)


def bls12_g2add(inp):
    if len(inp) != 512:
        raise EipError(INVALID_LENGTH)
    a = decode_g2(inp[:256])
    b = decode_g2(inp[256:])
    return encode_g2(g2_add(a, b))


def bls12_g2mul(inp):
    if len(inp) != 288:
        raise EipError(INVALID_LENGTH)
    a = decode_g2(inp[:256])
    k = decode_scalar(inp[256:])
    return encode_g2(g2_mul(a, k))


def bls12_g2multiexp(inp):
    if len(inp) == 0 or len(inp) % 288:
        raise EipError(INVALID_LENGTH)
    acc = None
    for o in range(0, len(inp), 288):
        a = decode_g2(inp[o:o + 256])
        k = decode_scalar(inp[o + 256:o + 288])
        acc = g2_add(acc, g2_mul(a, k))
    return encode_g2(acc)


def bls12_pairing(inp):
    """Order of checks per pair: G1 decode, G1 subgroup, G2 decode, G2 subgroup
    (src/eip2537.c:1033-1053); one final exponentiation (:1070)."""
    if len(inp) == 0 or len(inp) % 384:
        raise EipError(INVALID_LENGTH)
    f = F12_ONE
    for o in range(0, len(inp), 384):
        a = decode_g1(inp[o:o + 128])
        if not g1_in_subgroup(a):
            raise EipError(NOT_IN_SUBGROUP)
        b = decode_g2(inp[o + 128:o + 384])
        if not g2_in_subgroup(b):
            raise EipError(NOT_IN_SUBGROUP)
        f = f12_mul(f, miller_loop(a, b))
    ok = final_exp(f) == F12_ONE
    return bytes(31) + (b"\x01" if ok else b"\x00")


def call(fn, inp):
    """Run a precompile the way the C-ABI reports it: (code, out_bytes|None)."""
    try:
        return SUCCESS, fn(bytes(inp))
    except EipError as e:
        return e.code, None


# ----------------------------------------------------------------------------
# Deterministic input construction shared by tests / golden generation
# ----------------------------------------------------------------------------
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def scalar256(self):
        """Four big-endian 64-bit outputs, uniform on [0, 2^256), not reduced."""
        v = 0
        for _ in range(4):
            v = (v << 64) | self.next()
        return v


def random_g1(rng, in_subgroup=True):
    while True:
        x = rng.scalar256() * rng.scalar256() % P
        y = fp_sqrt((x * x * x + 4) % P)
        if y is None:
            continue
        pt = (x, y if rng.next() & 1 else (-y) % P)
        if in_subgroup:
            pt = g1_mul(pt, H1)
            if pt is None:
                continue
        return pt


def random_g2(rng, in_subgroup=True):
    while True:
        x = (rng.scalar256() * rng.scalar256() % P, rng.scalar256() * rng.scalar256() % P)
        y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), B2))
        if y is None:
            continue
        pt = (x, y if rng.next() & 1 else f2_neg(y))
        if in_subgroup:
            pt = g2_mul(pt, H2)
            if pt is None:
                continue
        return pt
