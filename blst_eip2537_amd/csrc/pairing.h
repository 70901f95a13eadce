// Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v), the optimal-ate Miller loop for
// BLS12-381 and the final exponentiation.  Replaces blst_miller_loop / blst_fp12_mul /
// blst_final_exp / blst_fp12_is_one as used by the reference at src/eip2537.c:1060-1076.
#pragma once
#include "curve.h"

namespace eip {

struct Fp6 { Fp2 a0, a1, a2; };
struct Fp12 { Fp6 c0, c1; };

HD Fp6 fp6_zero() { return Fp6{fp2_zero(), fp2_zero(), fp2_zero()}; }
HD Fp6 fp6_one() { return Fp6{fp2_one(), fp2_zero(), fp2_zero()}; }
HD Fp6 add(const Fp6 &a, const Fp6 &b) { return Fp6{add(a.a0, b.a0), add(a.a1, b.a1), add(a.a2, b.a2)}; }
HD Fp6 sub(const Fp6 &a, const Fp6 &b) { return Fp6{sub(a.a0, b.a0), sub(a.a1, b.a1), sub(a.a2, b.a2)}; }
HD Fp6 neg(const Fp6 &a) { return Fp6{neg(a.a0), neg(a.a1), neg(a.a2)}; }
HD bool eq(const Fp6 &a, const Fp6 &b) { return eq(a.a0, b.a0) && eq(a.a1, b.a1) && eq(a.a2, b.a2); }
HD Fp6 mul_by_v(const Fp6 &a) { return Fp6{mul_xi(a.a2), a.a0, a.a1}; }
HD Fp6 mul(const Fp6 &a, const Fp6 &b) {
    Fp2 v0 = mul(a.a0, b.a0), v1 = mul(a.a1, b.a1), v2 = mul(a.a2, b.a2);
    Fp2 c0 = add(v0, mul_xi(sub(sub(mul(add(a.a1, a.a2), add(b.a1, b.a2)), v1), v2)));
    Fp2 c1 = add(sub(sub(mul(add(a.a0, a.a1), add(b.a0, b.a1)), v0), v1), mul_xi(v2));
    Fp2 c2 = add(sub(sub(mul(add(a.a0, a.a2), add(b.a0, b.a2)), v0), v2), v1);
    return Fp6{c0, c1, c2};
}
// a * (b0 + b1 v)
HD Fp6 mul_by_01(const Fp6 &a, const Fp2 &b0, const Fp2 &b1) {
    return Fp6{add(mul(a.a0, b0), mul_xi(mul(a.a2, b1))),
               add(mul(a.a0, b1), mul(a.a1, b0)),
               add(mul(a.a1, b1), mul(a.a2, b0))};
}
// a * (b1 v)
HD Fp6 mul_by_1(const Fp6 &a, const Fp2 &b1) {
    return Fp6{mul_xi(mul(a.a2, b1)), mul(a.a0, b1), mul(a.a1, b1)};
}
HD Fp6 inv(const Fp6 &a) {
    Fp2 c0 = sub(sqr(a.a0), mul_xi(mul(a.a1, a.a2)));
    Fp2 c1 = sub(mul_xi(sqr(a.a2)), mul(a.a0, a.a1));
    Fp2 c2 = sub(sqr(a.a1), mul(a.a0, a.a2));
    Fp2 t = add(mul(a.a0, c0), mul_xi(add(mul(a.a2, c1), mul(a.a1, c2))));
    t = inv(t);
    return Fp6{mul(c0, t), mul(c1, t), mul(c2, t)};
}

HD Fp12 fp12_one() { return Fp12{fp6_one(), fp6_zero()}; }
HD bool eq(const Fp12 &a, const Fp12 &b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
HD bool is_one(const Fp12 &a) { return eq(a, fp12_one()); }
HD Fp12 conj(const Fp12 &a) { return Fp12{a.c0, neg(a.c1)}; }
HD Fp12 mul(const Fp12 &a, const Fp12 &b) {
    Fp6 t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    Fp6 c1 = sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1);
    return Fp12{add(t0, mul_by_v(t1)), c1};
}
HD Fp12 sqr(const Fp12 &a) {
    Fp6 t = mul(a.c0, a.c1);
    Fp6 s = mul(add(a.c0, a.c1), add(a.c0, mul_by_v(a.c1)));
    return Fp12{sub(sub(s, t), mul_by_v(t)), add(t, t)};
}
HD Fp12 inv(const Fp12 &a) {
    Fp6 t = inv(sub(mul(a.c0, a.c0), mul_by_v(mul(a.c1, a.c1))));
    return Fp12{mul(a.c0, t), neg(mul(a.c1, t))};
}
// f * ((a0 + a1 v) + (a4 v) w): the shape of every Miller-loop line
HD Fp12 mul_by_014(const Fp12 &f, const Fp2 &a0, const Fp2 &a1, const Fp2 &a4) {
    Fp6 r0 = add(mul_by_01(f.c0, a0, a1), mul_by_v(mul_by_1(f.c1, a4)));
    Fp6 r1 = add(mul_by_1(f.c0, a4), mul_by_01(f.c1, a0, a1));
    return Fp12{r0, r1};
}
HD Fp2 k_fp2(const Fp &c0, const Fp &c1) { return Fp2{c0, c1}; }
HD Fp12 frob(const Fp12 &a) {
    Fp12 r;
    r.c0.a0 = conj(a.c0.a0);
    r.c0.a1 = mul(conj(a.c0.a1), k_fp2(Fp{{K_FROB1_1_C0}}, Fp{{K_FROB1_1_C1}}));
    r.c0.a2 = mul(conj(a.c0.a2), k_fp2(Fp{{K_FROB1_2_C0}}, Fp{{K_FROB1_2_C1}}));
    r.c1.a0 = mul(conj(a.c1.a0), k_fp2(Fp{{K_FROB1_3_C0}}, Fp{{K_FROB1_3_C1}}));
    r.c1.a1 = mul(conj(a.c1.a1), k_fp2(Fp{{K_FROB1_4_C0}}, Fp{{K_FROB1_4_C1}}));
    r.c1.a2 = mul(conj(a.c1.a2), k_fp2(Fp{{K_FROB1_5_C0}}, Fp{{K_FROB1_5_C1}}));
    return r;
}
HD Fp12 frob2(const Fp12 &a) {
    Fp12 r;
    r.c0.a0 = a.c0.a0;
    r.c0.a1 = mul_fp(a.c0.a1, Fp{{K_FROB2_1}});
    r.c0.a2 = mul_fp(a.c0.a2, Fp{{K_FROB2_2}});
    r.c1.a0 = mul_fp(a.c1.a0, Fp{{K_FROB2_3}});
    r.c1.a1 = mul_fp(a.c1.a1, Fp{{K_FROB2_4}});
    r.c1.a2 = mul_fp(a.c1.a2, Fp{{K_FROB2_5}});
    return r;
}

// ------------------------------------------------------------------ Miller loop
// Running point T on the twist in Jacobian coordinates (X, Y, Z).  Each step returns the line
// through T scaled by Fp2 factors (killed by the final exponentiation):
//     l = a0 + (a1 xP) v + (a4 yP) v w
struct MillerT { Fp2 x, y, z; };
struct Line { Fp2 a0, a1, a4; };

HD Line miller_dbl_step(MillerT &T) {
    Fp2 A = sqr(T.x), B = sqr(T.y), C = sqr(B);
    Fp2 D = dbl(sub(sub(sqr(add(T.x, B)), A), C));
    Fp2 E = add(dbl(A), A);
    Fp2 ZZ = sqr(T.z);
    Fp2 X3 = sub(sqr(E), dbl(D));
    Fp2 Y3 = sub(mul(E, sub(D, X3)), dbl(dbl(dbl(C))));
    Fp2 Z3 = dbl(mul(T.y, T.z));
    Line l;
    l.a0 = sub(mul(E, T.x), dbl(B));      // 3X^3 - 2Y^2
    l.a1 = neg(mul(E, ZZ));               // -3X^2 Z^2
    l.a4 = mul(Z3, ZZ);                   // 2YZ^3
    T.x = X3; T.y = Y3; T.z = Z3;
    return l;
}
HD Line miller_add_step(MillerT &T, const Aff<Fp2> &Q) {
    Fp2 ZZ = sqr(T.z);
    Fp2 U2 = mul(Q.x, ZZ);
    Fp2 S2 = mul(Q.y, mul(ZZ, T.z));
    Fp2 H = sub(U2, T.x);
    Fp2 th = sub(S2, T.y);
    Fp2 HH = sqr(H);
    Fp2 HHH = mul(HH, H);
    Fp2 V = mul(T.x, HH);
    Fp2 X3 = sub(sub(sqr(th), HHH), dbl(V));
    Fp2 Y3 = sub(mul(th, sub(V, X3)), mul(T.y, HHH));
    Fp2 Z3 = mul(T.z, H);
    Line l;
    l.a0 = sub(mul(th, Q.x), mul(Z3, Q.y));
    l.a1 = neg(th);
    l.a4 = Z3;
    T.x = X3; T.y = Y3; T.z = Z3;
    return l;
}
// f_{|z|,Q}(P), conjugated because z < 0.  A pair with either point at infinity contributes
// the identity (EIP-2537 semantics; DESIGN.md "documented choices").
HD Fp12 miller_loop(const Aff<Fp> &P, const Aff<Fp2> &Q) {
    Fp12 f = fp12_one();
    if (is_inf(P) || is_inf(Q)) return f;
    MillerT T{Q.x, Q.y, fp2_one()};
    const uint64_t z = K_Z_ABS;
    for (int i = 62; i >= 0; i--) {
        Line l = miller_dbl_step(T);
        f = mul_by_014(sqr(f), l.a0, mul_fp(l.a1, P.x), mul_fp(l.a4, P.y));
        if ((z >> i) & 1ull) {
            l = miller_add_step(T, Q);
            f = mul_by_014(f, l.a0, mul_fp(l.a1, P.x), mul_fp(l.a4, P.y));
        }
    }
    return conj(f);
}

// The product of the Miller functions of n pairs with ONE chain of squarings (host route of small
// pairing checks: 63 Fp12 squarings per call instead of per pair).  Pairs with a point at infinity are skipped.
inline Fp12 miller_loop_multi(const Aff<Fp> *P, const Aff<Fp2> *Q, size_t n) {
    constexpr size_t kMax = 64;
    MillerT T[kMax];
    size_t idx[kMax], m = 0;
    for (size_t i = 0; i < n && m < kMax; i++)
        if (!is_inf(P[i]) && !is_inf(Q[i])) { T[m] = MillerT{Q[i].x, Q[i].y, fp2_one()}; idx[m++] = i; }
    Fp12 f = fp12_one();
    if (m == 0) return f;
    const uint64_t z = K_Z_ABS;
    for (int i = 62; i >= 0; i--) {
        if (i != 62) f = sqr(f);                              // the first square is of one
        for (size_t j = 0; j < m; j++) {
            const Line l = miller_dbl_step(T[j]);
            f = mul_by_014(f, l.a0, mul_fp(l.a1, P[idx[j]].x), mul_fp(l.a4, P[idx[j]].y));
        }
        if ((z >> i) & 1ull)
            for (size_t j = 0; j < m; j++) {
                const Line l = miller_add_step(T[j], Q[idx[j]]);
                f = mul_by_014(f, l.a0, mul_fp(l.a1, P[idx[j]].x), mul_fp(l.a4, P[idx[j]].y));
            }
    }
    return conj(f);
}

// The batched form of the loop above (pairing.hip): with L_s the product over all pairs of their line at
// step s (68 steps: 63 doublings, 5 additions), the product of the pairs' Miller functions is
// (...((L_0)^2 L_1)^2 ...), squared before every doubling step, conjugated because z < 0.
HD Fp12 miller_product_from_steps(const Fp12 *L) {
    Fp12 F = fp12_one();
    const uint64_t z = K_Z_ABS;
    int si = 0;
    for (int bit = 62; bit >= 0; bit--) {
        F = mul(sqr(F), L[si++]);
        if ((z >> bit) & 1ull) F = mul(F, L[si++]);
    }
    return conj(F);
}

// ------------------------------------------------------------------ final exponentiation
// Squaring in the cyclotomic subgroup (Granger-Scott): three Fp4 squarings = 9 Fp2 squarings.
// Valid only after the easy part of the final exponentiation; proven equal to the generic
// square on such elements in oracle/pymodel/fastmodel.py (f12_cyclotomic_sqr).
HD void fp4_sqr(Fp2 &c0, Fp2 &c1, const Fp2 &a, const Fp2 &b) {
    Fp2 t0 = sqr(a), t1 = sqr(b);
    c0 = add(mul_xi(t1), t0);
    c1 = sub(sub(sqr(add(a, b)), t0), t1);
}
HD Fp12 cyclotomic_sqr(const Fp12 &f) {
    Fp2 z0 = f.c0.a0, z4 = f.c0.a1, z3 = f.c0.a2, z2 = f.c1.a0, z1 = f.c1.a1, z5 = f.c1.a2;
    Fp2 t0, t1, t2, t3;
    fp4_sqr(t0, t1, z0, z1);
    z0 = add(dbl(sub(t0, z0)), t0);
    z1 = add(dbl(add(t1, z1)), t1);
    fp4_sqr(t0, t1, z2, z3);
    fp4_sqr(t2, t3, z4, z5);
    z4 = add(dbl(sub(t0, z4)), t0);
    z5 = add(dbl(add(t1, z5)), t1);
    t0 = mul_xi(t3);
    z2 = add(dbl(add(t0, z2)), t0);
    z3 = add(dbl(sub(t2, z3)), t2);
    return Fp12{Fp6{z0, z4, z3}, Fp6{z2, z1, z5}};
}
// g^z for g in the cyclotomic subgroup (inverse = conjugate)
HD Fp12 exp_by_z(const Fp12 &g) {
    const uint64_t z = K_Z_ABS;
    Fp12 acc = g;
    for (int i = 62; i >= 0; i--) {
        acc = cyclotomic_sqr(acc);
        if ((z >> i) & 1ull) acc = mul(acc, g);
    }
    return conj(acc);
}
// f^(3 (p^12-1)/r) via the easy part and (z-1)^2 (z+p) (z^2+p^2-1) + 3.  The reference only
// observes "== 1" (src/eip2537.c:1076) and gcd(3, r) = 1.
HD Fp12 final_exp(const Fp12 &f) {
    Fp12 f1 = mul(conj(f), inv(f));
    Fp12 f2 = mul(frob2(f1), f1);
    Fp12 y0 = mul(exp_by_z(f2), conj(f2));
    Fp12 y1 = mul(exp_by_z(y0), conj(y0));
    Fp12 y2 = mul(exp_by_z(y1), frob(y1));
    Fp12 y3 = mul(mul(exp_by_z(exp_by_z(y2)), frob2(y2)), conj(y2));
    return mul(y3, mul(sqr(f2), f2));
}

}  // namespace eip
