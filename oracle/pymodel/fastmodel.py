"""Optimised-formula layer of the Python model (TEST INFRASTRUCTURE ONLY).

Each routine here is the big-integer statement of a formula that the C oracle
(oracle/c) and the HIP engine (blst_eip2537_amd/csrc) implement with Montgomery
limbs; tests/test_pymodel.py proves each one equal to the generic definition in
bls12_381.py.  It is also where the derived constants (Frobenius coefficients,
psi / phi endomorphism constants) are computed for tools/gen_constants.py.

Reference anchors: blst_miller_loop / blst_final_exp / blst_p1_affine_in_g1 /
blst_p2_affine_in_g2 call sites at src/eip2537.c:1041-1070.
"""
from bls12_381 import *   # noqa: F401,F403

# ---------------------------------------------------------------------------
# Frobenius constants
# ---------------------------------------------------------------------------
GAMMA_V1 = f2_pow(XI, (P - 1) // 3)          # v^p   = GAMMA_V1 * v
GAMMA_V2 = f2_sqr(GAMMA_V1)                  # v^2p  = GAMMA_V2 * v^2
GAMMA_W = f2_pow(XI, (P - 1) // 6)           # w^p   = GAMMA_W * w


def f6_frob(a):
    return (f2_conj(a[0]), f2_mul(f2_conj(a[1]), GAMMA_V1), f2_mul(f2_conj(a[2]), GAMMA_V2))


def f12_frob(a):
    c0 = f6_frob(a[0])
    c1 = f6_frob(a[1])
    return (c0, tuple(f2_mul(x, GAMMA_W) for x in c1))


def f12_frob2(a):
    return f12_frob(f12_frob(a))


# p^2-Frobenius constants: all lie in Fp.
GAMMA2_V1 = f2_mul(GAMMA_V1, f2_conj(GAMMA_V1))        # v^(p^2)  = GAMMA2_V1 * v
GAMMA2_V2 = f2_sqr(GAMMA2_V1)
GAMMA2_W = f2_mul(GAMMA_W, f2_conj(GAMMA_W))
assert GAMMA2_V1[1] == 0 and GAMMA2_W[1] == 0


# ---------------------------------------------------------------------------
# Sparse line multiplication:  f * (a0 + a1*v + a4*v*w)
# ---------------------------------------------------------------------------
def f6_mul_by_01(a, b0, b1):
    """a * (b0 + b1 v)."""
    a0, a1, a2 = a
    return (f2_add(f2_mul(a0, b0), f2_mul(XI, f2_mul(a2, b1))),
            f2_add(f2_mul(a0, b1), f2_mul(a1, b0)),
            f2_add(f2_mul(a1, b1), f2_mul(a2, b0)))


def f6_mul_by_1(a, b1):
    """a * (b1 v)."""
    a0, a1, a2 = a
    return (f2_mul(XI, f2_mul(a2, b1)), f2_mul(a0, b1), f2_mul(a1, b1))


def f12_mul_by_014(f, a0, a1, a4):
    """f * ((a0 + a1 v) + (a4 v) w)."""
    c0, c1 = f
    t0 = f6_mul_by_01(c0, a0, a1)
    t1 = f6_mul_by_1(c1, a4)
    r0 = f6_add(t0, f6_mul_by_v(t1))
    r1 = f6_add(f6_mul_by_1(c0, a4), f6_mul_by_01(c1, a0, a1))
    return (r0, r1)


# ---------------------------------------------------------------------------
# Projective Miller loop on the twist (Jacobian T over Fp2, affine Q, affine P)
#
#   line * w^3 * (Fp2 scalar) = a0 + (a1 * xP) v + (a4 * yP) v w
# doubling:  a0 = 3X^3 - 2Y^2,  a1 = -3X^2 Z^2,  a4 = 2YZ^3
# addition:  a0 = th*x2 - mu*y2, a1 = -th,       a4 = mu
#            th = y2 Z^3 - Y,    mu = Z (x2 Z^2 - X)
# ---------------------------------------------------------------------------
def _dbl_step(T):
    X, Y, Zc = T
    A = f2_sqr(X)
    B = f2_sqr(Y)
    C = f2_sqr(B)
    D = f2_sub(f2_sub(f2_sqr(f2_add(X, B)), A), C)
    D = f2_add(D, D)
    E = f2_add(f2_add(A, A), A)
    Fq = f2_sqr(E)
    ZZ = f2_sqr(Zc)
    X3 = f2_sub(Fq, f2_add(D, D))
    C8 = f2_muls(C, 8)
    Y3 = f2_sub(f2_mul(E, f2_sub(D, X3)), C8)
    Z3 = f2_mul(f2_add(Y, Y), Zc)
    a0 = f2_sub(f2_mul(E, X), f2_add(B, B))
    a1 = f2_neg(f2_mul(E, ZZ))
    a4 = f2_mul(Z3, ZZ)
    return (X3, Y3, Z3), (a0, a1, a4)


def _add_step(T, Q):
    X, Y, Zc = T
    x2, y2 = Q
    ZZ = f2_sqr(Zc)
    U2 = f2_mul(x2, ZZ)
    S2 = f2_mul(y2, f2_mul(ZZ, Zc))
    H = f2_sub(U2, X)
    th = f2_sub(S2, Y)
    HH = f2_sqr(H)
    HHH = f2_mul(HH, H)
    V = f2_mul(X, HH)
    X3 = f2_sub(f2_sub(f2_sqr(th), HHH), f2_add(V, V))
    Y3 = f2_sub(f2_mul(th, f2_sub(V, X3)), f2_mul(Y, HHH))
    Z3 = f2_mul(Zc, H)
    a0 = f2_sub(f2_mul(th, x2), f2_mul(Z3, y2))
    a1 = f2_neg(th)
    a4 = Z3
    return (X3, Y3, Z3), (a0, a1, a4)


def miller_loop_fast(p, q):
    if p is None or q is None:
        return F12_ONE
    xp, yp = p
    T = (q[0], q[1], F2_ONE)
    f = F12_ONE
    for i in range(Z_ABS.bit_length() - 2, -1, -1):
        T, (a0, a1, a4) = _dbl_step(T)
        f = f12_mul_by_014(f12_sqr(f), a0, f2_muls(a1, xp), f2_muls(a4, yp))
        if (Z_ABS >> i) & 1:
            T, (a0, a1, a4) = _add_step(T, q)
            f = f12_mul_by_014(f, a0, f2_muls(a1, xp), f2_muls(a4, yp))
    return f12_conj(f)


# ---------------------------------------------------------------------------
# Final exponentiation, easy part then the (z-1)^2 (z+p) (z^2+p^2-1) + 3 chain.
# Computes f^(3 (p^12-1)/r); 3 is coprime to r so "== 1" is unchanged.
# ---------------------------------------------------------------------------
def _exp_by_z(g):
    """g^z for g in the cyclotomic subgroup (inverse = conjugate), z < 0."""
    return f12_conj(f12_pow(g, Z_ABS))


def final_exp_fast(f):
    f1 = f12_mul(f12_conj(f), f12_inv(f))        # ^(p^6 - 1)
    f2 = f12_mul(f12_frob2(f1), f1)              # ^(p^2 + 1)
    y0 = f12_mul(_exp_by_z(f2), f12_conj(f2))    # ^(z - 1)
    y1 = f12_mul(_exp_by_z(y0), f12_conj(y0))    # ^(z - 1)^2
    y2 = f12_mul(_exp_by_z(y1), f12_frob(y1))    # ^(z + p)
    y3 = f12_mul(f12_mul(_exp_by_z(_exp_by_z(y2)), f12_frob2(y2)), f12_conj(y2))  # ^(z^2+p^2-1)
    return f12_mul(y3, f12_mul(f12_sqr(f2), f2))


# ---------------------------------------------------------------------------
# Endomorphisms and fast subgroup tests
# ---------------------------------------------------------------------------
# psi on E'(Fp2): untwist, p-Frobenius, twist:  (x,y) -> (conj(x) PSI_X, conj(y) PSI_Y)
PSI_X = f2_inv(f2_pow(XI, (P - 1) // 3))
PSI_Y = f2_inv(f2_pow(XI, (P - 1) // 2))


def g2_psi(q):
    if q is None:
        return None
    return (f2_mul(f2_conj(q[0]), PSI_X), f2_mul(f2_conj(q[1]), PSI_Y))


def g2_in_subgroup_fast(q):
    """Q in G2  <=>  psi(Q) == [z]Q   (z < 0: [z]Q = -[|z|]Q)."""
    if q is None:
        return True
    return g2_psi(q) == ec_neg(FP2, g2_mul(q, Z_ABS))


# phi on E(Fp): (x,y) -> (BETA x, y); BETA is the cube root of unity for which
# phi acts on G1 as multiplication by -z^2 (mod r).
def _find_beta():
    lam = (-Z * Z) % R
    target = g1_mul(G1, lam)
    # primitive cube roots of unity: roots of b^2 + b + 1
    s = fp_sqrt((-3) % P)
    inv2 = pow(2, -1, P)
    for b in (((-1 + s) * inv2) % P, ((-1 - s) * inv2) % P):
        if (b * G1_X % P, G1_Y) == target:
            return b
    raise AssertionError("no beta")


BETA = _find_beta()


def g1_phi(a):
    return None if a is None else (a[0] * BETA % P, a[1])


def g1_in_subgroup_fast(a):
    """P in G1  <=>  phi(P) == [-z^2]P, computed as -[|z|]([|z|]P)."""
    if a is None:
        return True
    t = g1_mul(g1_mul(a, Z_ABS), Z_ABS)
    return g1_phi(a) == ec_neg(FP, t)


# ---------------------------------------------------------------------------
# Cyclotomic squaring (Granger-Scott) for elements of the cyclotomic subgroup
# (after the easy part of the final exponentiation): 9 Fp2 squarings.
# ---------------------------------------------------------------------------
def _fp4_square(a, b):
    t0 = f2_sqr(a)
    t1 = f2_sqr(b)
    c0 = f2_add(f2_mul(t1, XI), t0)
    c1 = f2_sub(f2_sub(f2_sqr(f2_add(a, b)), t0), t1)
    return c0, c1


def f12_cyclotomic_sqr(f):
    (z0, z4, z3), (z2, z1, z5) = f
    t0, t1 = _fp4_square(z0, z1)
    z0 = f2_add(f2_muls(f2_sub(t0, z0), 2), t0)
    z1 = f2_add(f2_muls(f2_add(t1, z1), 2), t1)
    t0, t1 = _fp4_square(z2, z3)
    t2, t3 = _fp4_square(z4, z5)
    z4 = f2_add(f2_muls(f2_sub(t0, z4), 2), t0)
    z5 = f2_add(f2_muls(f2_add(t1, z5), 2), t1)
    t0 = f2_mul(t3, XI)
    z2 = f2_add(f2_muls(f2_add(t0, z2), 2), t0)
    z3 = f2_add(f2_muls(f2_sub(t2, z3), 2), t2)
    return ((z0, z4, z3), (z2, z1, z5))
