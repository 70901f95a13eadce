// RFC 9380 map_to_curve for BLS12-381 G1 / G2: simplified SWU onto the isogenous curve, then the
// 11- / 3-isogeny.  Host code of the engine (single-element precompiles, BASELINE config 1 class:
// no GPU).  Replaces blst_map_to_g1(out, u, NULL) / blst_map_to_g2(out, u, NULL) as called by
// the reference at src/eip2537.c:1113 and :1155; the caller clears the cofactor.
// The isogeny tables are derived (tools/derive_isogeny.py, Velu's formulas) and the construction
// is pinned by the RFC 9380 appendix J vectors (tests/test_h2c.py).
#pragma once
#include "curve.h"
#include "iso_constants.h"

namespace eip {

inline Fp ld_fp(const uint32_t *w) { Fp r; memcpy(r.l, w, 48); return r; }
inline Fp2 ld_fp2(const uint32_t (*w)[12]) { return Fp2{ld_fp(w[0]), ld_fp(w[1])}; }

// square roots (p = 3 mod 4); return false when the argument is not a square
inline bool h2c_sqrt(Fp &r, const Fp &a) {
    const uint32_t e[12] = {K_P_PLUS_1_DIV_4};
    Fp s = fp_pow(a, e, 12);
    bool ok = eq(sqr(s), a);
    r = s;
    return ok;
}
inline bool h2c_sqrt(Fp2 &r, const Fp2 &a) {
    Fp x0;
    if (is_zero(a.c1)) {
        if (h2c_sqrt(x0, a.c0)) { r = Fp2{x0, fp_zero()}; return true; }
        h2c_sqrt(x0, neg(a.c0));                       // -a0 is a square when a0 is not
        r = Fp2{fp_zero(), x0};
        return true;
    }
    Fp n;
    if (!h2c_sqrt(n, add(sqr(a.c0), sqr(a.c1)))) return false;
    const Fp half = inv(dbl(fp_one()));
    for (int k = 0; k < 2; k++) {
        Fp t = mul(add(a.c0, k == 0 ? n : neg(n)), half);
        if (!h2c_sqrt(x0, t) || is_zero(x0)) continue;
        Fp2 cand{x0, mul(a.c1, inv(dbl(x0)))};
        if (eq(sqr(cand), a)) { r = cand; return true; }
    }
    return false;
}
inline int h2c_sgn0(const Fp &a) { return (int)(fp_from_mont(a).l[0] & 1u); }
inline int h2c_sgn0(const Fp2 &a) {
    Fp r0 = fp_from_mont(a.c0), r1 = fp_from_mont(a.c1);
    return (int)((r0.l[0] & 1u) | ((is_zero(r0) ? 1u : 0u) & (r1.l[0] & 1u)));
}

template <class F> struct IsoTables;
template <> struct IsoTables<Fp> {
    static Fp A() { return ld_fp(K_ISO_G1_A); }
    static Fp B() { return ld_fp(K_ISO_G1_B); }
    static Fp Z() { return ld_fp(K_ISO_G1_Z); }
    static Fp MBA() { return ld_fp(K_ISO_G1_MBA); }
    static Fp BZA() { return ld_fp(K_ISO_G1_BZA); }
    static Fp xnum(int i) { return ld_fp(K_ISO_G1_XNUM[i]); }
    static Fp xden(int i) { return ld_fp(K_ISO_G1_XDEN[i]); }
    static Fp ynum(int i) { return ld_fp(K_ISO_G1_YNUM[i]); }
    static Fp yden(int i) { return ld_fp(K_ISO_G1_YDEN[i]); }
    static constexpr int kXnum = 12, kXden = 11, kYnum = 16, kYden = 16;
};
template <> struct IsoTables<Fp2> {
    static Fp2 A() { return ld_fp2(K_ISO_G2_A); }
    static Fp2 B() { return ld_fp2(K_ISO_G2_B); }
    static Fp2 Z() { return ld_fp2(K_ISO_G2_Z); }
    static Fp2 MBA() { return ld_fp2(K_ISO_G2_MBA); }
    static Fp2 BZA() { return ld_fp2(K_ISO_G2_BZA); }
    static Fp2 xnum(int i) { return ld_fp2(K_ISO_G2_XNUM[i]); }
    static Fp2 xden(int i) { return ld_fp2(K_ISO_G2_XDEN[i]); }
    static Fp2 ynum(int i) { return ld_fp2(K_ISO_G2_YNUM[i]); }
    static Fp2 yden(int i) { return ld_fp2(K_ISO_G2_YDEN[i]); }
    static constexpr int kXnum = 4, kXden = 3, kYnum = 4, kYden = 4;
};

template <class F, class Get> inline F h2c_horner(Get get, int n, const F &x) {
    F acc = get(n - 1);
    for (int i = n - 2; i >= 0; i--) acc = add(mul(acc, x), get(i));
    return acc;
}

// iso(sswu(u)) as an affine point of the target curve ((0,0) = infinity)
template <class F> inline Aff<F> map_to_curve(const F &u) {
    using T = IsoTables<F>;
    const F A = T::A(), B = T::B();
    F zu2 = mul(T::Z(), sqr(u));
    F tv1 = add(sqr(zu2), zu2);
    F x1 = is_zero(tv1) ? T::BZA() : mul(T::MBA(), add(f_one<F>(), inv(tv1)));
    F gx1 = add(mul(add(sqr(x1), A), x1), B);
    F x = x1, y;
    if (!h2c_sqrt(y, gx1)) {
        x = mul(zu2, x1);
        h2c_sqrt(y, add(mul(add(sqr(x), A), x), B));      // a square whenever gx1 is not
    }
    if (h2c_sgn0(u) != h2c_sgn0(y)) y = neg(y);
    F xd = h2c_horner<F>(T::xden, T::kXden, x), yd = h2c_horner<F>(T::yden, T::kYden, x);
    if (is_zero(xd) || is_zero(yd)) return Aff<F>{f_zero<F>(), f_zero<F>()};
    F xn = h2c_horner<F>(T::xnum, T::kXnum, x), yn = h2c_horner<F>(T::ynum, T::kYnum, x);
    return Aff<F>{mul(xn, inv(xd)), mul(y, mul(yn, inv(yd)))};
}

}  // namespace eip
