// Short-Weierstrass arithmetic for E: y^2 = x^3 + 4 over Fp (G1) and E': y^2 = x^3 + 4(1+u)
// over Fp2 (G2), in extended Jacobian ("XYZZ") coordinates:  x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2,
// infinity <=> ZZ = 0.  Every addition is complete (handles infinity, P = Q, P = -Q) because
// duplicates, inverses, infinity and non-subgroup points are all legal MSM inputs of the ABI
// (reference src/eip2537.c:331-340; blst_p1_add_or_double / _affine call sites :461,:605,:693).
#pragma once
#include "field.h"

namespace eip {

template <class F> struct Aff { F x, y; };          // (0,0) encodes infinity
template <class F> struct Xyzz { F x, y, zz, zzz; };

template <class F> HD bool is_inf(const Aff<F> &a) { return is_zero(a.x) && is_zero(a.y); }
template <class F> HD bool is_inf(const Xyzz<F> &p) { return is_zero(p.zz); }
template <class F> HD Xyzz<F> xyzz_inf() { return Xyzz<F>{f_zero<F>(), f_zero<F>(), f_zero<F>(), f_zero<F>()}; }
template <class F> HD Xyzz<F> from_affine(const Aff<F> &a) {
    if (is_inf(a)) return xyzz_inf<F>();
    return Xyzz<F>{a.x, a.y, f_one<F>(), f_one<F>()};
}
template <class F> HD Aff<F> neg(const Aff<F> &a) { return Aff<F>{a.x, neg(a.y)}; }
template <class F> HD Xyzz<F> neg(const Xyzz<F> &p) { return Xyzz<F>{p.x, neg(p.y), p.zz, p.zzz}; }

// y^2 == x^3 + b  (blst_p1_affine_on_curve / blst_p2_affine_on_curve, reference :336,:397)
template <class F> HD bool on_curve(const Aff<F> &a) {
    return eq(sqr(a.y), add(mul(sqr(a.x), a.x), curve_b<F>()));
}

// 2P for affine P != infinity (mdbl-2008-s-1)
template <class F> HD Xyzz<F> dbl_affine(const Aff<F> &a) {
    F U = dbl(a.y);
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(a.x, V);
    F XX = sqr(a.x);
    F M = add(dbl(XX), XX);
    F X3 = sub(sqr(M), dbl(S));
    F Y3 = sub(mul(M, sub(S, X3)), mul(W, a.y));
    return Xyzz<F>{X3, Y3, V, W};
}
// 2P (dbl-2008-s-1); infinity stays infinity because ZZ3 = V*ZZ
template <class F> HD Xyzz<F> dbl(const Xyzz<F> &p) {
    F U = dbl(p.y);
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(p.x, V);
    F XX = sqr(p.x);
    F M = add(dbl(XX), XX);
    F X3 = sub(sqr(M), dbl(S));
    F Y3 = sub(mul(M, sub(S, X3)), mul(W, p.y));
    return Xyzz<F>{X3, Y3, mul(V, p.zz), mul(W, p.zzz)};
}
// P + Q, Q affine (madd-2008-s), complete
template <class F> HD Xyzz<F> madd(const Xyzz<F> &p, const Aff<F> &q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return Xyzz<F>{q.x, q.y, f_one<F>(), f_one<F>()};
    F U2 = mul(q.x, p.zz);
    F S2 = mul(q.y, p.zzz);
    F Pd = sub(U2, p.x);
    F R = sub(S2, p.y);
    if (is_zero(Pd)) {
        if (is_zero(R)) return dbl_affine(q);
        return xyzz_inf<F>();
    }
    F PP = sqr(Pd);
    F PPP = mul(Pd, PP);
    F Q = mul(p.x, PP);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    F Y3 = sub(mul(R, sub(Q, X3)), mul(p.y, PPP));
    return Xyzz<F>{X3, Y3, mul(p.zz, PP), mul(p.zzz, PPP)};
}
// P + Q (add-2008-s), complete
template <class F> HD Xyzz<F> add(const Xyzz<F> &p, const Xyzz<F> &q) {
    if (is_inf(q)) return p;
    if (is_inf(p)) return q;
    F U1 = mul(p.x, q.zz);
    F U2 = mul(q.x, p.zz);
    F S1 = mul(p.y, q.zzz);
    F S2 = mul(q.y, p.zzz);
    F Pd = sub(U2, U1);
    F R = sub(S2, S1);
    if (is_zero(Pd)) {
        if (is_zero(R)) return dbl(p);
        return xyzz_inf<F>();
    }
    F PP = sqr(Pd);
    F PPP = mul(Pd, PP);
    F Q = mul(U1, PP);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    F Y3 = sub(mul(R, sub(Q, X3)), mul(S1, PPP));
    return Xyzz<F>{X3, Y3, mul(mul(p.zz, q.zz), PP), mul(mul(p.zzz, q.zzz), PPP)};
}
// one inversion: 1/(ZZ*ZZZ); infinity -> (0,0)
template <class F> HD Aff<F> to_affine(const Xyzz<F> &p) {
    F t = inv(mul(p.zz, p.zzz));
    F izz = mul(t, p.zzz);
    F izzz = mul(t, p.zz);
    return Aff<F>{mul(p.x, izz), mul(p.y, izzz)};
}
// does the projective point equal the affine point a (a != infinity)?
template <class F> HD bool eq_affine(const Xyzz<F> &p, const Aff<F> &a) {
    if (is_inf(p)) return false;
    return eq(mul(a.x, p.zz), p.x) && eq(mul(a.y, p.zzz), p.y);
}

// [k]P for the low nbits bits of a little-endian 32-bit-word scalar: true multiplication on
// the whole curve -- no reduction mod r and no endomorphism, because mul/MSM inputs are not
// subgroup-checked (reference src/eip2537.c:340,401; blst_p1_mult call sites :514,:602).
template <class F> HD Xyzz<F> scalar_mul(const Aff<F> &a, const uint32_t *k, int nbits) {
    Xyzz<F> acc = xyzz_inf<F>();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = madd(acc, a);
    }
    return acc;
}
// [m]P for a small unsigned m and projective P (bucket-reduce offsets)
template <class F> HD Xyzz<F> small_mul(const Xyzz<F> &p, uint32_t m) {
    if (m == 0) return xyzz_inf<F>();
    Xyzz<F> acc = p;                                 // the top bit
    for (int i = 30 - __builtin_clz(m); i >= 0; i--) {
        acc = dbl(acc);
        if ((m >> i) & 1u) acc = add(acc, p);
    }
    return acc;
}
// [|z|]P, |z| = 0xd201000000010000 (bits 63,62,60,57,48,16)
template <class F> HD Xyzz<F> mul_zabs(const Aff<F> &a) {
    const uint64_t z = K_Z_ABS;
    Xyzz<F> acc = from_affine(a);
    for (int i = 62; i >= 0; i--) {
        acc = dbl(acc);
        if ((z >> i) & 1ull) acc = madd(acc, a);
    }
    return acc;
}
template <class F> HD Xyzz<F> mul_zabs(const Xyzz<F> &p) {
    const uint64_t z = K_Z_ABS;
    Xyzz<F> acc = p;
    for (int i = 62; i >= 0; i--) {
        acc = dbl(acc);
        if ((z >> i) & 1ull) acc = add(acc, p);
    }
    return acc;
}

// Exact r-torsion membership (blst_p1_affine_in_g1 / blst_p2_affine_in_g2, reference :1041,:1051).
// G1: phi(P) == -[z^2]P with phi(x,y) = (beta x, y);  G2: psi(Q) == [z]Q = -[|z|]Q.
// Both forms are checked against [r]P == infinity on every prime-order torsion subgroup of the
// two cofactors in tests/test_oracle.py.
HD bool in_g1(const Aff<Fp> &a) {
    if (is_inf(a)) return true;
    Xyzz<Fp> t = mul_zabs(mul_zabs(a));
    Aff<Fp> phi_neg{mul(a.x, Fp{{K_BETA}}), neg(a.y)};
    return eq_affine(t, phi_neg);
}
HD bool in_g2(const Aff<Fp2> &a) {
    if (is_inf(a)) return true;
    Xyzz<Fp2> t = mul_zabs(a);
    Aff<Fp2> psi_neg{mul(conj(a.x), Fp2{Fp{{K_PSI_X_C0}}, Fp{{K_PSI_X_C1}}}),
                     neg(mul(conj(a.y), Fp2{Fp{{K_PSI_Y_C0}}, Fp{{K_PSI_Y_C1}}}))};
    return eq_affine(t, psi_neg);
}

// n doublings in a row of a host accumulator: AVX-512 IFMA vectors where they pay (ifma_horner.h, which every host translation unit
// that calls msm_interleaved() includes), the loop over dbl() otherwise
template <class F> inline void horner_double_n(Xyzz<F> &acc, int n);
// in_g2() for the host route of small pairing checks: the same test, the runs of doublings of |z| as chains (63 doublings in six runs)
inline bool in_g2_host(const Aff<Fp2> &a) {
    if (is_inf(a)) return true;
    Xyzz<Fp2> t = from_affine(a);
    static_assert(K_Z_ABS == ((1ull << 63) | (1ull << 62) | (1ull << 60) | (1ull << 57) | (1ull << 48) | (1ull << 16)), "bits of |z|");
    int prev = 63;
    for (int bit : {62, 60, 57, 48, 16}) {
        horner_double_n(t, prev - bit);
        t = madd(t, a);
        prev = bit;
    }
    horner_double_n(t, prev);
    Aff<Fp2> psi_neg{mul(conj(a.x), Fp2{Fp{{K_PSI_X_C0}}, Fp{{K_PSI_X_C1}}}),
                     neg(mul(conj(a.y), Fp2{Fp{{K_PSI_Y_C0}}, Fp{{K_PSI_Y_C1}}}))};
    return eq_affine(t, psi_neg);
}
// sum k_i P_i for a few points on the host: interleaved signed 5-bit windows (Straus).  k: n x 8 little-endian
// words (256-bit, NOT reduced: the points need not lie in the prime-order subgroup).  Signed digits in
// [-15, 16] with k = sum d_w 32^w over 52 windows (the last one takes the final carry), a table of the
// multiples 1 .. 16 per point, one chain of 5 doublings per window shared by all points.
// nwords: 32-bit words per scalar (8: the 256-bit scalars of the precompiles; the 636-bit cofactor of the G2 map takes 20).
template <class F> inline Xyzz<F> msm_interleaved(const Aff<F> *pts, const uint32_t *k, size_t n, int nwords = 8) {
    constexpr int kW = 5, kTable = 1 << (kW - 1);
    const int kWindows = (32 * nwords + kW) / kW;             // one more than the bits need: the last window takes the final carry
    Xyzz<F> *table = new Xyzz<F>[n * kTable];
    signed char *digits = new signed char[n * kWindows];
    for (size_t i = 0; i < n; i++) {
        const uint32_t *ki = k + i * (size_t)nwords;
        int carry = 0;
        for (int w = 0; w < kWindows; w++) {
            const int bit = w * kW, word = bit >> 5, sh = bit & 31;
            uint32_t v = word < nwords ? ki[word] >> sh : 0u;
            if (sh > 32 - kW && word + 1 < nwords) v |= ki[word + 1] << (32 - sh);
            int d = (int)(v & ((1u << kW) - 1u)) + carry;
            carry = d > kTable;
            if (carry) d -= 1 << kW;
            digits[i * kWindows + w] = (signed char)d;
        }
        Xyzz<F> *t = table + i * kTable;
        t[0] = from_affine(pts[i]);                         // infinity stays infinity in every entry
        t[1] = dbl(t[0]);
        for (int m = 2; m < kTable; m++) t[m] = madd(t[m - 1], pts[i]);
    }
    Xyzz<F> acc = xyzz_inf<F>();
    for (int w = kWindows - 1; w >= 0; w--) {
        if (!is_inf(acc)) horner_double_n(acc, kW);
        for (size_t i = 0; i < n; i++) {
            const int d = digits[i * kWindows + w];
            if (d > 0) acc = add(acc, table[i * kTable + d - 1]);
            else if (d < 0) acc = add(acc, neg(table[i * kTable - d - 1]));
        }
    }
    delete[] table;
    delete[] digits;
    return acc;
}

}  // namespace eip
