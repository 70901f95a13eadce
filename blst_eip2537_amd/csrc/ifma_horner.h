// AVX-512 IFMA doubling chains for the host tail of a multiexp.
//
// After the device pipeline the host combines the W window sums by Horner's rule (msm.hip): c doublings per window, 256 in all
// whatever the input size -- G1: ~0.3 us each on the scalar host code, G2: ~1 us, i.e. 0.08 / 0.25 ms of EVERY call (a 2^16-record
// G2 multiexp is 1.6 ms in all, a 128-record one 0.7).  A doubling is nine products in three dependent rounds; here the rounds run as
// vector products of ifma.h (eight Fp values per vector, 52-bit limbs, factor 2^416, lazy bounds), a whole G2 point in ONE vector:
//     lanes 0,1 = X   2,3 = Y   4,5 = ZZ   6,7 = ZZZ      (Fp2 = lane pair: component 0, component 1)
// and four Fp2 products at a time on the lane pairs (fp2_mul4: two accumulating 8-lane products, one reduction).  G1 takes one lane per
// coordinate and plain vector products.  The additions between the chains stay scalar code (curve.h: complete, with the equal /
// opposite / infinite cases) -- two conversions per window.
//
// dbl-2008-s-1, as curve.h's dbl():   U = 2 Y, V = U^2, W = U V, S = X V, M = 3 X^2,
//     X3 = M^2 - 2 S,  Y3 = M (S - X3) - W Y,  ZZ3 = V ZZ,  ZZZ3 = W ZZZ.      Infinity (ZZ = 0) stays infinity.
// Bounds (units of p) between doublings: X3, Y3 < 66 (a product below 2 plus the 64 p of a difference), ZZ3, ZZZ3 < 2; every
// operand of a product stays far below the 2^15 p that two accumulated products allow, and second operands below the 512 p that
// fp2_mul4 subtracts them from.  tools/ifma_check.hip compares chains of doublings with curve.h on random and special points.
#pragma once
#include "ifma.h"
#if defined(EIP_HAVE_IFMA)
#include "curve.h"

namespace eip {
namespace ifma {

// four Fp2 products on the lane pairs: out pair j = A pair j * B pair j.  A, B normalised; B <= 512 p.
EIP_IFMA V8 fp2_mul4(const V8 &A, const V8 &B) {
    const __m512i dup_even = _mm512_setr_epi64(0, 0, 2, 2, 4, 4, 6, 6), dup_odd = _mm512_setr_epi64(1, 1, 3, 3, 5, 5, 7, 7);
    const __m512i swap = _mm512_setr_epi64(1, 0, 3, 2, 5, 4, 7, 6);
    const V8 a0 = vperm(A, dup_even), a1 = vperm(A, dup_odd);
    const V8 bs = vperm(B, swap);                                           // (b1, b0)
    const V8 q = vblend(0xAA, vnorm(vsub512(vzero(), bs)), bs);             // even lanes: 512 p - b1, odd lanes: b0
    __m512i t[17];
    for (int j = 0; j < 17; j++) t[j] = _mm512_setzero_si512();
    vmac(t, a0, B);                                                         // a0 b0            | a0 b1
    vmac(t, a1, q);                                                         // a1 (512 p - b1)  | a1 b0
    return vredc(t);
}
// 2 P for a G2 point in one vector (lane pairs X, Y, ZZ, ZZZ); input normalised with the bounds above
EIP_IFMA V8 g2_dbl(const V8 &P) {
    const V8 D = vnorm(vblend(0x0C, P, vdbl(P)));                           // [X, U = 2 Y, ZZ, ZZZ]
    const V8 R1 = fp2_mul4(D, D);                                           // [XX, V, -, -]
    const V8 T = vnorm(vblend(0x03, R1, vtriple(R1)));                      // [M = 3 XX, V, -, -]
    const __m512i ia2 = _mm512_setr_epi64(2, 3, 0, 1, 8 + 0, 8 + 1, 8 + 2, 8 + 3);      // from D | T:  [U, X, M, V]
    const __m512i ib2 = _mm512_setr_epi64(2, 3, 2, 3, 0, 1, 8 + 4, 8 + 5);              // from T | D:  [V, V, M, ZZ]
    const V8 R2 = fp2_mul4(vperm2(D, ia2, T), vperm2(T, ib2, D));           // [W, S, MM, ZZ3]
    const __m512i all_mm = _mm512_setr_epi64(4, 5, 4, 5, 4, 5, 4, 5), all_s = _mm512_setr_epi64(2, 3, 2, 3, 2, 3, 2, 3);
    const V8 Sv = vperm(R2, all_s);
    const V8 X3 = vnorm(vsub64(vperm(R2, all_mm), vdbl(Sv)));               // M^2 - 2 S in every pair, < 66
    const V8 Dif = vnorm(vsub512(Sv, X3));                                  // S - X3, < 514: first operand below
    const __m512i ia3 = _mm512_setr_epi64(0, 1, 8 + 0, 8 + 1, 8 + 0, 8 + 1, 8 + 0, 8 + 1);          // from Dif | R2:  [S - X3, W, W, W]
    const __m512i ib3 = _mm512_setr_epi64(0, 1, 8 + 2, 8 + 3, 8 + 6, 8 + 7, 8 + 6, 8 + 7);          // from T | P:     [M, Y, ZZZ, ZZZ]
    const V8 R3 = fp2_mul4(vperm2(Dif, ia3, R2), vperm2(T, ib3, P));        // [M (S - X3), W Y, ZZZ3, -]
    const __m512i t1_to_0 = _mm512_setr_epi64(2, 3, 2, 3, 2, 3, 2, 3), t0_all = _mm512_setr_epi64(0, 1, 0, 1, 0, 1, 0, 1);
    const V8 Y3 = vsub64(vperm(R3, t0_all), vperm(R3, t1_to_0));            // in every pair, < 66
    const __m512i iz = _mm512_setr_epi64(0, 1, 0, 1, 6, 7, 8 + 4, 8 + 5);   // from R2 | R3:  [-, -, ZZ3, ZZZ3]
    const V8 Z = vperm2(R2, iz, R3);
    return vnorm(vblend(0x03, vblend(0x0C, Z, Y3), X3));                    // [X3, Y3, ZZ3, ZZZ3]
}
// the same for G1: lane 0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ (lanes 4 .. 7 carry copies that nothing reads)
EIP_IFMA V8 g1_dbl(const V8 &P) {
    const V8 D = vnorm(vblend(0x02, P, vdbl(P)));                           // [X, U, ZZ, ZZZ]
    const V8 R1 = vmul(D, D);                                               // [XX, V, -, -]
    const V8 T = vnorm(vblend(0x01, R1, vtriple(R1)));                      // [M, V, -, -]
    const __m512i ia2 = _mm512_setr_epi64(1, 0, 8 + 0, 8 + 1, 1, 0, 8 + 0, 8 + 1);      // from D | T:  [U, X, M, V]
    const __m512i ib2 = _mm512_setr_epi64(1, 1, 0, 8 + 2, 1, 1, 0, 8 + 2);              // from T | D:  [V, V, M, ZZ]
    const V8 R2 = vmul(vperm2(D, ia2, T), vperm2(T, ib2, D));               // [W, S, MM, ZZ3]
    const __m512i all_mm = _mm512_set1_epi64(2), all_s = _mm512_set1_epi64(1);
    const V8 Sv = vperm(R2, all_s);
    const V8 X3 = vnorm(vsub64(vperm(R2, all_mm), vdbl(Sv)));
    const V8 Dif = vnorm(vsub512(Sv, X3));
    const __m512i ia3 = _mm512_setr_epi64(0, 8 + 0, 8 + 0, 8 + 0, 0, 8 + 0, 8 + 0, 8 + 0);          // from Dif | R2:  [S - X3, W, W, W]
    const __m512i ib3 = _mm512_setr_epi64(0, 8 + 1, 8 + 3, 8 + 3, 0, 8 + 1, 8 + 3, 8 + 3);          // from T | P:     [M, Y, ZZZ, ZZZ]
    const V8 R3 = vmul(vperm2(Dif, ia3, R2), vperm2(T, ib3, P));            // [M (S - X3), W Y, ZZZ3, -]
    const V8 Y3 = vsub64(vperm(R3, _mm512_set1_epi64(0)), vperm(R3, _mm512_set1_epi64(1)));
    const __m512i iz = _mm512_setr_epi64(0, 0, 3, 8 + 2, 0, 0, 3, 8 + 2);   // from R2 | R3:  [-, -, ZZ3, ZZZ3]
    const V8 Z = vperm2(R2, iz, R3);
    return vnorm(vblend(0x11, vblend(0x22, Z, Y3), X3));
}
// lanes whose value is 0 mod p (input normalised, below 2^20 p)
EIP_IFMA __mmask8 lanes_zero_modp(const V8 &a) {
    const V8 r = vreduce(a);                                                // < 3 p, normalised: 0, p or 2 p if it is a multiple
    const Consts &k = consts();
    __mmask8 z = 0xff, one = 0xff, two = 0xff;
    uint64_t carry = 0;
    for (int j = 0; j < 8; j++) {
        const uint64_t d = 2 * k.p[j] + carry;                              // limbs of 2 p
        z &= _mm512_cmpeq_epi64_mask(r.l[j], _mm512_setzero_si512());
        one &= _mm512_cmpeq_epi64_mask(r.l[j], _mm512_set1_epi64((long long)k.p[j]));
        two &= _mm512_cmpeq_epi64_mask(r.l[j], _mm512_set1_epi64((long long)(j < 7 ? d & kMask52 : d)));
        carry = d >> 52;
    }
    return (__mmask8)(z | one | two);
}
// P1 + P2 (add-2008-s) for two G2 points that are not infinity; `special` is raised (and the result is to be ignored) when the two have
// the same x -- equal or opposite points: the caller then takes the complete scalar addition.  Inputs normalised, X, Y < 2^9 p.
EIP_IFMA V8 g2_add(const V8 &P1, const V8 &P2, bool &special) {
    const __m512i ia1 = _mm512_setr_epi64(0, 1, 8 + 0, 8 + 1, 2, 3, 8 + 2, 8 + 3);                  // P1 | P2:  [X1, X2, Y1, Y2]
    const __m512i ib1 = _mm512_setr_epi64(8 + 4, 8 + 5, 4, 5, 8 + 6, 8 + 7, 6, 7);                  // P1 | P2:  [ZZ2, ZZ1, ZZZ2, ZZZ1]
    const V8 R1 = fp2_mul4(vperm2(P1, ia1, P2), vperm2(P1, ib1, P2));       // [U1, U2, S1, S2]
    const __m512i swap_pairs = _mm512_setr_epi64(2, 3, 0, 1, 6, 7, 4, 5);
    const V8 Dm = vnorm(vsub64(vperm(R1, swap_pairs), R1));                 // pair 0: P = U2 - U1, pair 2: R = S2 - S1   (< 66)
    special = (lanes_zero_modp(Dm) & 0x03) == 0x03;
    const __m512i ia2 = _mm512_setr_epi64(0, 1, 4, 5, 8 + 4, 8 + 5, 8 + 6, 8 + 7);                  // Dm | P1:  [P, R, ZZ1, ZZZ1]
    const V8 R2 = fp2_mul4(vperm2(Dm, ia2, P1), vperm2(Dm, ia2, P2));       // [PP, RR, ZZ12, ZZZ12]   (second: [P, R, ZZ2, ZZZ2])
    const __m512i it3 = _mm512_setr_epi64(0, 1, 0, 1, 8 + 4, 8 + 5, 8 + 4, 8 + 5);                  // R1 | R2:  [U1, U1, ZZ12, ZZ12]
    const __m512i ia3 = _mm512_setr_epi64(0, 1, 8 + 0, 8 + 1, 8 + 4, 8 + 5, 8 + 4, 8 + 5);          // Dm | T3:  [P, U1, ZZ12, ZZ12]
    const __m512i pp_all = _mm512_setr_epi64(0, 1, 0, 1, 0, 1, 0, 1);
    const V8 R3 = fp2_mul4(vperm2(Dm, ia3, vperm2(R1, it3, R2)), vperm(R2, pp_all));                // [PPP, Q, ZZ3, -]
    const __m512i pair1_all = _mm512_setr_epi64(2, 3, 2, 3, 2, 3, 2, 3);
    const V8 Qa = vperm(R3, pair1_all), PPPa = vperm(R3, pp_all);
    const V8 X3 = vnorm(vsub64(vsub64(vperm(R2, pair1_all), PPPa), vdbl(Qa)));                      // RR - PPP - 2 Q, < 130
    const V8 Dif = vnorm(vsub512(Qa, X3));                                  // Q - X3, < 514: first operand
    const __m512i it4 = _mm512_setr_epi64(0, 1, 4, 5, 8 + 6, 8 + 7, 8 + 6, 8 + 7);                  // R1 | R2:  [-, S1, ZZZ12, ZZZ12]
    const V8 A4 = vblend(0x03, vperm2(R1, it4, R2), Dif);                   // [Q - X3, S1, ZZZ12, ZZZ12]
    const __m512i ib4 = _mm512_setr_epi64(4, 5, 8 + 0, 8 + 1, 8 + 0, 8 + 1, 8 + 0, 8 + 1);          // Dm | R3:  [R, PPP, PPP, PPP]
    const V8 R4 = fp2_mul4(A4, vperm2(Dm, ib4, R3));                        // [t0, t1, ZZZ3, -]
    const V8 Y3 = vsub64(vperm(R4, pp_all), vperm(R4, pair1_all));          // in every pair, < 66
    const __m512i iz = _mm512_setr_epi64(0, 1, 0, 1, 4, 5, 8 + 4, 8 + 5);   // R3 | R4:  [-, -, ZZ3, ZZZ3]
    return vnorm(vblend(0x03, vblend(0x0C, vperm2(R3, iz, R4), Y3), X3));   // [X3, Y3, ZZ3, ZZZ3]
}
EIP_IFMA V8 g2_load(const Xyzz<Fp2> &a) {
    const Fp *src[8] = {&a.x.c0, &a.x.c1, &a.y.c0, &a.y.c1, &a.zz.c0, &a.zz.c1, &a.zzz.c0, &a.zzz.c1};
    return vload(src);
}
EIP_IFMA Xyzz<Fp2> g2_store(const V8 &v) {
    Xyzz<Fp2> a;
    Fp *dst[8] = {&a.x.c0, &a.x.c1, &a.y.c0, &a.y.c1, &a.zz.c0, &a.zz.c1, &a.zzz.c0, &a.zzz.c1};
    vstore(dst, v);
    return a;
}
// the Horner accumulator of a G2 multiexp kept in ONE vector from the first window sum to the last: doubling chains and additions
// without conversions in between (each window sum is converted once on its way in)
struct G2Horner {
    V8 v;
    bool inf = true;
    EIP_IFMA void dbl_n(int n) { if (!inf) for (int i = 0; i < n; i++) v = g2_dbl(v); }
    EIP_IFMA void add(const Xyzz<Fp2> &q) {
        if (is_zero(q.zz)) return;
        if (inf) { v = g2_load(q); inf = false; return; }
        bool special;
        const V8 r = g2_add(v, g2_load(q), special);
        if (!special) { v = r; return; }
        const Xyzz<Fp2> a = eip::add(g2_store(v), q);                      // equal or opposite points: the complete scalar addition
        if (is_zero(a.zz)) inf = true; else v = g2_load(a);
    }
    EIP_IFMA Xyzz<Fp2> result() const { return inf ? xyzz_inf<Fp2>() : g2_store(v); }
};
// acc <- 2^n acc
EIP_IFMA void double_n(Xyzz<Fp2> &acc, int n) {
    const Fp *src[8] = {&acc.x.c0, &acc.x.c1, &acc.y.c0, &acc.y.c1, &acc.zz.c0, &acc.zz.c1, &acc.zzz.c0, &acc.zzz.c1};
    V8 v = vload(src);
    for (int i = 0; i < n; i++) v = g2_dbl(v);
    Fp *dst[8] = {&acc.x.c0, &acc.x.c1, &acc.y.c0, &acc.y.c1, &acc.zz.c0, &acc.zz.c1, &acc.zzz.c0, &acc.zzz.c1};
    vstore(dst, v);
}
EIP_IFMA void double_n(Xyzz<Fp> &acc, int n) {
    const Fp *src[8] = {&acc.x, &acc.y, &acc.zz, &acc.zzz, nullptr, nullptr, nullptr, nullptr};
    V8 v = vload(src);
    for (int i = 0; i < n; i++) v = g1_dbl(v);
    Fp *dst[8] = {&acc.x, &acc.y, &acc.zz, &acc.zzz, nullptr, nullptr, nullptr, nullptr};
    vstore(dst, v);
}

}  // namespace ifma

// n doublings of the Horner accumulator: IFMA vectors where the CPU has them (and EIP2537_HOST_IFMA is not 0), scalar otherwise.
// G1 keeps the scalar code by default: a G1 doubling fills three of a vector product's eight lanes, and the mulx / adx product of
// field.h is as fast (measured: tools/ifma_check.hip prints both; EIP2537_HOST_IFMA_G1=1 selects the vectors).
inline bool horner_vectors(const Fp2 *) { return host_ifma_enabled(); }
inline bool horner_vectors(const Fp *) {
    static const bool on = [] { const char *v = getenv("EIP2537_HOST_IFMA_G1"); return v && atoi(v) != 0; }();
    return on && host_ifma_enabled();
}
template <class F> inline void horner_double_n(Xyzz<F> &acc, int n) {
    if (n > 0 && !is_zero(acc.zz) && horner_vectors((const F *)nullptr)) { ifma::double_n(acc, n); return; }
    for (int i = 0; i < n; i++) acc = dbl(acc);
}
// The accumulator of a Horner pass: dbl_n / add / result.  G2 with IFMA: one vector throughout (G2Horner); otherwise the scalar point.
template <class F> struct HornerAcc {
    Xyzz<F> acc = xyzz_inf<F>();
    void dbl_n(int n) { horner_double_n(acc, n); }
    void add(const Xyzz<F> &q) { acc = eip::add(acc, q); }
    Xyzz<F> result() const { return acc; }
};
template <> struct HornerAcc<Fp2> {
    Xyzz<Fp2> acc = xyzz_inf<Fp2>();
    ifma::G2Horner vec;
    const bool use_vec = host_ifma_enabled();
    void dbl_n(int n) { if (use_vec) vec.dbl_n(n); else horner_double_n(acc, n); }
    void add(const Xyzz<Fp2> &q) { if (use_vec) vec.add(q); else acc = eip::add(acc, q); }
    Xyzz<Fp2> result() const { return use_vec ? vec.result() : acc; }
};

}  // namespace eip
#else
#include "curve.h"
namespace eip {
template <class F> inline void horner_double_n(Xyzz<F> &acc, int n) { for (int i = 0; i < n; i++) acc = dbl(acc); }
template <class F> struct HornerAcc {
    Xyzz<F> acc = xyzz_inf<F>();
    void dbl_n(int n) { horner_double_n(acc, n); }
    void add(const Xyzz<F> &q) { acc = eip::add(acc, q); }
    Xyzz<F> result() const { return acc; }
};
}  // namespace eip
#endif
