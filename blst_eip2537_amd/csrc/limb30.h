// "Limb form" of Fp for long chains of products: the one-lane G1 accumulate (msm.hip, k_msm_accum_l).
//
// The column product of field.h spends about 40 % of its instructions outside the multiply-adds:
// re-slicing both 12 x 32-bit operands into 13 x 30-bit limbs, cutting the result back at bit 24 and
// conditional corrections in the additions between the products.  A kernel that keeps its values in
// limbs from the first product to the last pays for none of that:
//   * FpL holds 13 limbs of 30 bits (the top one takes what is left), value < 25 p;
//   * the Montgomery factor is R' = 2^390 = 13 x 30 bits, so the reduced product IS columns 13 .. 25,
//     and a product of a b < 630 p^2 (= R' p) comes out below 2 p;
//   * differences are a + K p - b with a constant K p >= b and one signed carry pass: no comparison,
//     no conditional subtraction; the bound of every value is tracked by hand where it is used.
// Values enter through the decode kernel (which writes x R', y R', -y R' of every point in limbs) and
// leave through to_fpi(), which multiplies by 2^384 mod p (back to the factor R of every other kernel)
// and packs 12 x 32-bit words.  All of it is HD: tools/limb30_check.hip runs the same code on the host
// against the 64-bit host product.
#pragma once
#include "curve.h"

namespace eip {

struct FpL { uint32_t l[13]; };
static constexpr uint32_t kM30 = 0x3fffffffu;

// K p in limbs
template <int K> HD void kp30(uint32_t out[13]) {
    const uint32_t p30[13] = {K_P30};
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const uint64_t v = (uint64_t)p30[k] * (uint32_t)K + carry;
        out[k] = k < 12 ? (uint32_t)v & kM30 : (uint32_t)v;
        carry = v >> 30;
    }
}

HD FpL fpl_zero() { FpL r; for (int k = 0; k < 13; k++) r.l[k] = 0; return r; }
HD FpL fpl_one() { return FpL{{K_R390_MODP_30}}; }              // 1 * R' mod p

// 12 x 32-bit words (value < 2^384) <-> limbs, value unchanged
HD FpL to_limbs(const Fp &a) {
    FpL r;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int bit = 30 * k, i = bit >> 5, s = bit & 31;
        uint32_t v = a.l[i] >> s;
        if (s > 2 && i + 1 < 12) v |= a.l[i + 1] << (32 - s);
        r.l[k] = v & kM30;
    }
    return r;
}
HD Fp from_limbs(const FpL &a) {
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, q = bit / 30, o = bit % 30;
        uint64_t t = (uint64_t)a.l[q] | ((uint64_t)a.l[q + 1] << 30) | (q + 2 < 13 ? (uint64_t)a.l[q + 2] << 60 : 0);
        r.l[w] = (uint32_t)(t >> o);
    }
    return r;
}

// a + K p - b, for b <= K p.  Result < a + K p.
template <int K> HD FpL subL(const FpL &a, const FpL &b) {
    uint32_t kp[13];
    kp30<K>(kp);
    FpL r;
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int32_t t = (int32_t)(a.l[k] + kp[k] - b.l[k]) + c;       // in (-2^30, 2^31)
        r.l[k] = (uint32_t)t & kM30;
        c = t >> 30;
    }
    r.l[12] = a.l[12] + kp[12] - b.l[12] + (uint32_t)c;
    return r;
}
// a + K p - 2 b, for 2 b <= K p.
template <int K> HD FpL sub2L(const FpL &a, const FpL &b) {
    uint32_t kp[13];
    kp30<K>(kp);
    FpL r;
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int32_t t = (int32_t)(a.l[k] + kp[k] - 2u * b.l[k]) + c;  // in [-2^31, 2^31)
        r.l[k] = (uint32_t)t & kM30;
        c = t >> 30;
    }
    r.l[12] = a.l[12] + kp[12] - 2u * b.l[12] + (uint32_t)c;
    return r;
}
// a + b and 2 a + b (no reduction: the value grows)
HD FpL addL(const FpL &a, const FpL &b) {
    FpL r;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint32_t t = a.l[k] + b.l[k] + c;
        r.l[k] = t & kM30;
        c = t >> 30;
    }
    r.l[12] = a.l[12] + b.l[12] + c;
    return r;
}
HD FpL dbl_addL(const FpL &a, const FpL &b) {
    FpL r;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const uint32_t t = 2u * a.l[k] + b.l[k] + c;                    // < 3 * 2^30 + 4
        r.l[k] = t & kM30;
        c = t >> 30;
    }
    r.l[12] = 2u * a.l[12] + b.l[12] + c;
    return r;
}

// a value below bound * p that is a multiple of p is j p with j = a[0] p^-1 mod 2^30 < bound: one
// multiplication decides all but 2^-30 * bound of the cases
HD bool is_zero_modp(const FpL &a, uint32_t bound) {
    const uint32_t j = (a.l[0] * K_PINV_30) & kM30;
    if (j >= bound) return false;
    const uint32_t p30[13] = {K_P30};
    uint64_t carry = 0;
    bool same = true;
#pragma unroll                                          // a rolled loop indexes a.l[] dynamically: the caller's value would live in scratch
    for (int k = 0; k < 13; k++) {
        const uint64_t v = (uint64_t)p30[k] * j + carry;
        same &= (k < 12 ? (uint32_t)v & kM30 : (uint32_t)v) == a.l[k];
        carry = v >> 30;
    }
    return same;
}

// Columns 13 .. 25 after the 13 reduction steps, normalised
HD FpL fpl_take_high(const uint64_t *col) {
    FpL r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const uint64_t v = col[13 + k] + carry;
        r.l[k] = k < 12 ? (uint32_t)v & kM30 : (uint32_t)v;
        carry = v >> 30;
    }
    return r;
}
// Column overflow.  Every limb is < 2^30, so a product term is <= (2^30 - 1)^2 and 16 of them fit a 64-bit column; a
// column of the 13 x 13 product takes up to 13, and the reduction adds m p_j terms on top (with the real limbs of p: at most
// 6.9 x 2^60 per column).  tools/limb_column_bounds.py tracks the exact worst case of every column through the schedules
// below and finds the fewest carry-outs that keep all of them below 2^64: columns 10 .. 14 once for a product or a square
// (round 2 carried 8 .. 16), and for the two-product sum 6 .. 18 after row 7 of both products plus column 12 before the
// reduction (round 2: 8 .. 16 and 4 .. 20).  A carry-out moves only the HIGH DWORD (x 4 into the next column, 2^32 =
// 4 x 2^30), which is one multiply-add and a clear instead of a 64-bit shift, a 64-bit add and a mask; the column keeps its
// low 32 bits (col_carry_hi, field.h -- the 12 x 32-bit products there follow the same schedules).
// a b / 2^390 + (< p), for a b < 630 p^2.
HD FpL mulL(const FpL &a, const FpL &b) {
    const uint32_t p30[13] = {K_P30};
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t m = ((uint32_t)col[i] * K_N0_30) & kM30;
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        col[i + 1] += col[i] >> 30;
        if (i == 7) {
#pragma unroll
            for (int c = 10; c <= 14; c++) col_carry_hi(col, c);
        }
    }
    return fpl_take_high(col);
}
HD FpL sqrL(const FpL &a) {
    const uint32_t p30[13] = {K_P30};
    uint32_t a2[13];
#pragma unroll
    for (int k = 0; k < 13; k++) a2[k] = a.l[k] << 1;
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        col[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < 13; j++) col[i + j] += (uint64_t)a2[j] * a.l[i];
    }
    // at most 7 terms of < 2^61 per column so far; the reduction adds 13 more
#pragma unroll
    for (int c = 10; c <= 14; c++) col_carry_hi(col, c);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t m = ((uint32_t)col[i] * K_N0_30) & kM30;
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        col[i + 1] += col[i] >> 30;
    }
    return fpl_take_high(col);
}

// (a b + c d) / 2^390 + (< p) with ONE reduction, for a b + c d < 630 p^2: 169 multiply-adds and a carry
// pass less than two products and an addition.  The rows of the two products alternate; after row 7 of both a column
// holds at most 16 terms.
HD FpL mul2L(const FpL &a, const FpL &b, const FpL &c, const FpL &d) {
    const uint32_t p30[13] = {K_P30};
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)c.l[j] * d.l[i];
        if (i == 7) {
#pragma unroll
            for (int k = 6; k <= 18; k++) col_carry_hi(col, k);
        }
    }
    col_carry_hi(col, 12);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t m = ((uint32_t)col[i] * K_N0_30) & kM30;
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        col[i + 1] += col[i] >> 30;
    }
    return fpl_take_high(col);
}
// a0 b0 + a1 b1 + ... + a5 b5 (three pairs of products) / 2^390 + (< p) with ONE reduction (round 4: the line products of the pairing fold
// add three two-product sums per coefficient; 2 x 169 multiply-adds and two carry / normalise passes less than three mul2L and two additions).
// Schedule found and checked by tools/limb_column_bounds.py (worst column 0.99999999 x 2^64 on all-ones limbs): the rows of a pair alternate as
// in mul2L with its carry-out of columns 6..18 after row 7; BETWEEN pairs the high dwords of columns 2..22 move up; column 12 before the reduction.
HD void mul_pair_rows(uint64_t col[27], const FpL &a, const FpL &b, const FpL &c, const FpL &d) {
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)c.l[j] * d.l[i];
        if (i == 7) {
#pragma unroll
            for (int k = 6; k <= 18; k++) col_carry_hi(col, k);
        }
    }
}
HD FpL mul6L(const FpL &a0, const FpL &b0, const FpL &a1, const FpL &b1, const FpL &a2, const FpL &b2,
             const FpL &a3, const FpL &b3, const FpL &a4, const FpL &b4, const FpL &a5, const FpL &b5) {
    const uint32_t p30[13] = {K_P30};
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
    mul_pair_rows(col, a0, b0, a1, b1);
#pragma unroll
    for (int k = 2; k <= 22; k++) col_carry_hi(col, k);
    mul_pair_rows(col, a2, b2, a3, b3);
#pragma unroll
    for (int k = 2; k <= 22; k++) col_carry_hi(col, k);
    mul_pair_rows(col, a4, b4, a5, b5);
    col_carry_hi(col, 12);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t m = ((uint32_t)col[i] * K_N0_30) & kM30;
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        col[i + 1] += col[i] >> 30;
    }
    return fpl_take_high(col);
}
// a0 b0 + a1 b1 + a2 b2 + a3 b3 (two pairs) / 2^390 + (< p) with ONE reduction: the first two pairs of mul6L's schedule, so every column
// stays below what tools/limb_column_bounds.py found for the six-product sum (all terms are non-negative).  The G2 accumulate's
// Y3 = R (Q - X3) - Y1 PPP over Fp2 is one of these per component (msm.hip, k_msm_accum2c_l).
HD FpL mul4L(const FpL &a0, const FpL &b0, const FpL &a1, const FpL &b1, const FpL &a2, const FpL &b2, const FpL &a3, const FpL &b3) {
    const uint32_t p30[13] = {K_P30};
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
    mul_pair_rows(col, a0, b0, a1, b1);
#pragma unroll
    for (int k = 2; k <= 22; k++) col_carry_hi(col, k);
    mul_pair_rows(col, a2, b2, a3, b3);
    col_carry_hi(col, 12);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const uint32_t m = ((uint32_t)col[i] * K_N0_30) & kM30;
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        col[i + 1] += col[i] >> 30;
    }
    return fpl_take_high(col);
}
// K p - b, for b <= K p
template <int K> HD FpL negL(const FpL &b) {
    uint32_t kp[13];
    kp30<K>(kp);
    FpL r;
    int32_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int32_t t = (int32_t)(kp[k] - b.l[k]) + c;
        r.l[k] = (uint32_t)t & kM30;
        c = t >> 30;
    }
    r.l[12] = kp[12] - b.l[12] + (uint32_t)c;
    return r;
}

// x R (12 x 32, canonical, Montgomery factor R = 2^384) -> x R' in limbs, through one product of the R world
HD FpL fpl_from_mont(const Fp &a) { return to_limbs(mul(a, Fp{{K_R390_MODP}})); }
// x R' (< 600 p) -> x R in [0, 2p) as 12 x 32-bit words
HD FpI to_fpi(const FpL &a) { return FpI{from_limbs(mulL(a, FpL{{K_R384_MODP_30}}))}; }

// ---- mixed addition on limb-form values -------------------------------------------------------------
// The formulas of madd() / dbl_affine() (curve.h).  Bounds held by the accumulator between entries, in
// units of p:  x < 8, y < 4, zz < 2, zzz < 2;  inside: P < 10, R < 6, every product of a b < 630.
struct AccL { FpL x, y, zz, zzz; };
// 2Q for an affine Q != infinity (mdbl-2008-s-1); rare (an entry equal to the running sum)
HD AccL dbl_affine_l(const FpL &qx, const FpL &qy) {
    const FpL U = addL(qy, qy);                       // < 2
    const FpL V = sqrL(U), W = mulL(U, V), S = mulL(qx, V), XX = sqrL(qx);
    const FpL M = dbl_addL(XX, XX);                   // 3 x^2 < 6
    const FpL X3 = sub2L<4>(sqrL(M), S);              // < 6
    const FpL Y3 = subL<2>(mulL(M, subL<6>(S, X3)), mulL(W, qy));    // S - X3 + 6p < 8; result < 4
    return AccL{X3, Y3, V, W};
}
// acc += (qx, qy), an affine point that is not infinity; inf says that acc is the point at infinity
HD void madd_l(AccL &acc, bool &inf, const FpL &qx, const FpL &qy) {
    if (inf) {
        acc = AccL{qx, qy, fpl_one(), fpl_one()};
        inf = false;
        return;
    }
    const FpL P = subL<8>(mulL(qx, acc.zz), acc.x);               // < 10
    const FpL R = subL<4>(mulL(qy, acc.zzz), acc.y);              // < 6
    if (is_zero_modp(P, 10)) {
        if (is_zero_modp(R, 6)) acc = dbl_affine_l(qx, qy);
        else inf = true;
        return;
    }
    const FpL PP = sqrL(P), PPP = mulL(P, PP), Q = mulL(acc.x, PP);
    const FpL X3 = sub2L<4>(subL<2>(sqrL(R), PPP), Q);            // R^2 - PPP - 2Q + 6p < 8
    acc.zz = mulL(acc.zz, PP);
    acc.zzz = mulL(acc.zzz, PPP);
    acc.y = mul2L(R, subL<8>(Q, X3), acc.y, negL<2>(PPP));        // R (Q - X3) - Y1 PPP: 6 x 10 + 4 x 2 < 630; result < 2
    acc.x = X3;
}

// ---- complete XYZZ operations on limb-form points (fold and one-lane reduce kernels) ----------------
// "Standard" bounds of a stored point, in units of p: x < 8, y < 4, zz < 2, zzz < 2 (what madd_l, add and
// dbl below all produce).  The point at infinity is zz = 0 in all limbs: a computed zz is a product of
// values that are not multiples of p and is never 0 mod p, and 0 * anything stays exactly 0 in mulL.
template <> HD FpL f_zero<FpL>() { return fpl_zero(); }
template <> HD FpL f_one<FpL>() { return fpl_one(); }
HD bool is_zero(const FpL &a) {                       // exact zero only (the infinity marker)
    uint32_t z = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) z |= a.l[k];
    return z == 0;
}
// 2P (dbl-2008-s-1); infinity stays infinity because ZZ3 = V * ZZ
__host__ __device__ __forceinline__ Xyzz<FpL> dbl(const Xyzz<FpL> &p) {
    const FpL U = addL(p.y, p.y);                                 // < 8
    const FpL V = sqrL(U), W = mulL(U, V), S = mulL(p.x, V), XX = sqrL(p.x);     // 64, 16, 16, 64 < 630
    const FpL M = dbl_addL(XX, XX);                               // < 6
    const FpL X3 = sub2L<4>(sqrL(M), S);                          // < 6
    const FpL Y3 = mul2L(M, subL<6>(S, X3), W, negL<4>(p.y));     // M (S - X3) - W Y1: 6 x 8 + 2 x 4; < 2
    return Xyzz<FpL>{X3, Y3, mulL(V, p.zz), mulL(W, p.zzz)};
}
// P + Q (add-2008-s), complete
__host__ __device__ __forceinline__ Xyzz<FpL> add(const Xyzz<FpL> &p, const Xyzz<FpL> &q) {
    if (is_zero(q.zz)) return p;
    if (is_zero(p.zz)) return q;
    const FpL U1 = mulL(p.x, q.zz), U2 = mulL(q.x, p.zz), S1 = mulL(p.y, q.zzz), S2 = mulL(q.y, p.zzz);   // 16, 16, 8, 8
    const FpL P = subL<2>(U2, U1), R = subL<2>(S2, S1);           // < 4
    if (is_zero_modp(P, 4)) {
        if (is_zero_modp(R, 4)) return dbl(p);
        return Xyzz<FpL>{fpl_zero(), fpl_zero(), fpl_zero(), fpl_zero()};
    }
    const FpL PP = sqrL(P), PPP = mulL(P, PP), Q = mulL(U1, PP);
    const FpL X3 = sub2L<4>(subL<2>(sqrL(R), PPP), Q);            // < 8
    const FpL Y3 = mul2L(R, subL<8>(Q, X3), S1, negL<2>(PPP));    // R (Q - X3) - S1 PPP: 4 x 10 + 2 x 2; < 2
    return Xyzz<FpL>{X3, Y3, mulL(mulL(p.zz, q.zz), PP), mulL(mulL(p.zzz, q.zzz), PPP)};
}
// back to the canonical 12 x 32-bit form of every other kernel and of the host
HD Xyzz<Fp> canon(const Xyzz<FpL> &p) {
    return Xyzz<Fp>{fp_reduce_once(to_fpi(p.x).v), fp_reduce_once(to_fpi(p.y).v), fp_reduce_once(to_fpi(p.zz).v),
                    fp_reduce_once(to_fpi(p.zzz).v)};
}

}  // namespace eip
