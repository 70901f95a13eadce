// AVX-512 IFMA arithmetic for the host tail of a pairing check: eight independent Fp values per vector, 52-bit limbs.
//
// After the device pipeline a check still owes the reference's single final exponentiation (src/eip2537.c:1070), whose
// hard part is 5 x 63 cyclotomic squarings in a row on one host core -- 0.26 of the 0.5 ms the host spent per 2^12-pair
// check in round 3 (scalar code: 18 products of 25 ns and ~45 additions per squaring).  A squaring is nine independent
// Fp2 squarings, i.e. eighteen independent Fp products: here they run eight per instruction on vpmadd52{lo,hi}uq.
//
//   V8            8 lanes x 8 limbs of 52 bits; lane = one Fp value in Montgomery form with R = 2^416 (the scalar host code
//                 uses R = 2^384: conversion is one product by a constant each way).  Values are kept lazily:
//                 R / p = 2^35, so a product of operands below 2^16 p comes out below 2 p without any correction;
//                 sums just add, a difference is a + k p - b, and the running state is brought back below 3 p
//                 once per squaring by a weak reduction (quotient estimate from the top limb).
//   limbs         a product reads only the low 52 bits of its operand limbs, so operands are carry-normalised first.
//
// Compiled into the library unconditionally (function-level target attributes); used when the CPU reports AVX-512 IFMA
// (the GPU box's host does: AMD EPYC 9575F), otherwise the scalar code of pairing.h runs.  tools/ifma_check.hip compares
// both on random and extreme inputs.
#pragma once
#include "pairing.h"
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define EIP_HAVE_IFMA 1
#include <immintrin.h>
#include <stdlib.h>
#include "pairing.h"

namespace eip {
namespace ifma {

#define EIP_IFMA __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl,avx512bw"))) inline

struct V8 { __m512i l[8]; };
static constexpr uint64_t kMask52 = (1ull << 52) - 1;

inline bool cpu_has_ifma() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq");
    return ok;
}

// ---- constants (little-endian 52-bit limbs), computed once from the 32-bit word tables ----------------------------------
struct Consts {
    uint64_t p[8];          // p
    uint64_t n0;            // -p^-1 mod 2^52
    uint64_t to_v[8];       // 2^448 mod p: (x 2^384) * this / 2^416 = x 2^416
    uint64_t from_v[8];     // 2^384 mod p as a plain integer: (x 2^416) * this / 2^416 = x 2^384
    uint64_t kp[8];         // 64 p in "borrow-proof" limbs (every limb but the top carries 2^56 extra, paid back by the next)
    uint64_t kp512[8];      // 512 p, the same way
    double inv_ptop;        // 1 / (floor(p / 2^364) + 1)
};
inline void words_to_limbs52(uint64_t out[8], const uint32_t w[12]) {
    unsigned __int128 acc = 0;
    int bits = 0, wi = 0;
    for (int j = 0; j < 8; j++) {
        while (bits < 52 && wi < 12) { acc |= (unsigned __int128)w[wi++] << bits; bits += 32; }
        out[j] = (uint64_t)acc & kMask52;
        acc >>= 52;
        bits -= 52;
        if (bits < 0) bits = 0;
    }
}
inline const Consts &consts() {
    static const Consts c = [] {
        Consts k{};
        const uint32_t pw[12] = {K_P};
        words_to_limbs52(k.p, pw);
        uint64_t inv = 1;                                      // Newton: p^-1 mod 2^64, then negate and cut
        for (int i = 0; i < 6; i++) inv *= 2 - k.p[0] * inv;
        k.n0 = (0 - inv) & kMask52;
        // 2^e mod p by repeated doubling of the scalar-domain one (fp_one() = 2^384 mod p as a plain integer)
        auto pow2_mod_p = [](int e) {                          // e >= 384
            Fp v = fp_one();
            for (int i = 384; i < e; i++) v = add(v, v);
            return v;
        };
        words_to_limbs52(k.to_v, pow2_mod_p(448).l);
        words_to_limbs52(k.from_v, pow2_mod_p(384).l);
        auto multiple = [&](uint64_t out[8], uint64_t times) {
            unsigned __int128 carry = 0;
            uint64_t m[8];
            for (int j = 0; j < 8; j++) {
                const unsigned __int128 v = (unsigned __int128)k.p[j] * times + carry;
                m[j] = (uint64_t)v & kMask52;
                carry = v >> 52;
            }
            for (int j = 0; j < 8; j++) out[j] = m[j] + (j < 7 ? (1ull << 56) : 0) - (j > 0 ? (1ull << 4) : 0);
        };
        multiple(k.kp, 64);
        multiple(k.kp512, 512);
        k.inv_ptop = 1.0 / (double)(k.p[7] + 1);
        return k;
    }();
    return c;
}

EIP_IFMA V8 vbroadcast(const uint64_t w[8]) {
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_set1_epi64((long long)w[j]);
    return r;
}
EIP_IFMA V8 vzero() {
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_setzero_si512();
    return r;
}
// carry-normalise: limbs 0 .. 6 below 2^52 (signed carries: limbs may be negative after a subtraction)
EIP_IFMA V8 vnorm(const V8 &a) {
    const __m512i mask = _mm512_set1_epi64((long long)kMask52);
    V8 r;
    __m512i c = _mm512_setzero_si512();
    for (int j = 0; j < 7; j++) {
        const __m512i t = _mm512_add_epi64(a.l[j], c);
        r.l[j] = _mm512_and_si512(t, mask);
        c = _mm512_srai_epi64(t, 52);
    }
    r.l[7] = _mm512_add_epi64(a.l[7], c);
    return r;
}
EIP_IFMA V8 vadd(const V8 &a, const V8 &b) {
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_add_epi64(a.l[j], b.l[j]);
    return r;
}
EIP_IFMA V8 vsub_raw(const V8 &a, const V8 &b) {             // limbs may go negative: normalise (signed) before any product
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_sub_epi64(a.l[j], b.l[j]);
    return r;
}
// a + 64 p - b, for b <= 64 p with limbs below 2^56
EIP_IFMA V8 vsub64(const V8 &a, const V8 &b) {
    const Consts &k = consts();
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_sub_epi64(_mm512_add_epi64(a.l[j], _mm512_set1_epi64((long long)k.kp[j])), b.l[j]);
    return r;
}
EIP_IFMA V8 vsub512(const V8 &a, const V8 &b) {             // a + 512 p - b, for b <= 512 p
    const Consts &k = consts();
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_sub_epi64(_mm512_add_epi64(a.l[j], _mm512_set1_epi64((long long)k.kp512[j])), b.l[j]);
    return r;
}
EIP_IFMA V8 vdbl(const V8 &a) { return vadd(a, a); }
EIP_IFMA V8 vtriple(const V8 &a) { return vadd(vadd(a, a), a); }
template <class Idx> EIP_IFMA V8 vperm(const V8 &a, Idx idx) {
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_permutexvar_epi64(idx, a.l[j]);
    return r;
}
EIP_IFMA V8 vperm2(const V8 &a, __m512i idx, const V8 &b) {   // lane i <- (idx[i] < 8 ? a : b)[idx[i] & 7]
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_permutex2var_epi64(a.l[j], idx, b.l[j]);
    return r;
}
EIP_IFMA V8 vblend(__mmask8 m, const V8 &a, const V8 &b) {    // lane i <- m[i] ? b : a
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = _mm512_mask_blend_epi64(m, a.l[j], b.l[j]);
    return r;
}

// t += a b as 16 columns of 52-bit partial products (no reduction): up to ~200 of these fit a 64-bit column
EIP_IFMA void vmac(__m512i t[17], const V8 &a, const V8 &b) {
    for (int i = 0; i < 8; i++) {
        const __m512i bi = b.l[i];
        for (int j = 0; j < 8; j++) {
            t[i + j] = _mm512_madd52lo_epu64(t[i + j], a.l[j], bi);
            t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], a.l[j], bi);
        }
    }
}
// Montgomery reduction of accumulated columns: (sum) / 2^416 + (less than p), normalised
EIP_IFMA V8 vredc(__m512i t[17]) {
    const Consts &k = consts();
    const __m512i n0 = _mm512_set1_epi64((long long)k.n0), zero = _mm512_setzero_si512();
    __m512i pv[8];
    for (int j = 0; j < 8; j++) pv[j] = _mm512_set1_epi64((long long)k.p[j]);
    for (int i = 0; i < 8; i++) {
        const __m512i m = _mm512_madd52lo_epu64(zero, t[i], n0);          // low 52 bits of t[i] * n0
        for (int j = 0; j < 8; j++) {
            t[i + j] = _mm512_madd52lo_epu64(t[i + j], m, pv[j]);
            t[i + j + 1] = _mm512_madd52hi_epu64(t[i + j + 1], m, pv[j]);
        }
        t[i + 1] = _mm512_add_epi64(t[i + 1], _mm512_srli_epi64(t[i], 52));
    }
    V8 r;
    for (int j = 0; j < 8; j++) r.l[j] = t[8 + j];
    return vnorm(r);
}
// Montgomery product a b / 2^416 (+ less than p), operands carry-normalised with a b < 2^32 p^2; result normalised, below 2 p
EIP_IFMA V8 vmul(const V8 &a, const V8 &b) {
    __m512i t[17];
    for (int j = 0; j < 17; j++) t[j] = _mm512_setzero_si512();
    vmac(t, a, b);
    return vredc(t);
}
// weak reduction: a normalised value below 2^20 p comes back congruent, normalised and below 3 p
EIP_IFMA V8 vreduce(const V8 &a) {
    const Consts &k = consts();
    const __m512d qd = _mm512_mul_pd(_mm512_cvtepu64_pd(a.l[7]), _mm512_set1_pd(k.inv_ptop));
    const __m512i q = _mm512_cvttpd_epu64(qd);                              // <= floor(top / (ptop + 1)): never too large
    const __m512i zero = _mm512_setzero_si512();
    V8 qp;
    __m512i hi = zero;
    for (int j = 0; j < 8; j++) {
        const __m512i pj = _mm512_set1_epi64((long long)k.p[j]);
        qp.l[j] = _mm512_add_epi64(_mm512_madd52lo_epu64(zero, q, pj), hi);
        hi = _mm512_madd52hi_epu64(zero, q, pj);
    }
    return vnorm(vsub_raw(a, qp));
}

// ---- conversions -------------------------------------------------------------------------------------------------------
// eight Fp values of the scalar host code (12 x 32-bit words, factor 2^384, canonical) -> one vector (factor 2^416)
EIP_IFMA V8 vload(const Fp *const src[8]) {
    alignas(64) uint64_t st[8][8];
    for (int lane = 0; lane < 8; lane++) {
        uint64_t limbs[8];
        static const Fp zero_fp = fp_zero();
        words_to_limbs52(limbs, src[lane] ? src[lane]->l : zero_fp.l);
        for (int j = 0; j < 8; j++) st[j][lane] = limbs[j];
    }
    V8 v;
    for (int j = 0; j < 8; j++) v.l[j] = _mm512_load_si512(st[j]);
    return vmul(v, vbroadcast(consts().to_v));
}
// ... and back: canonical scalar values (dst[lane] may be null)
EIP_IFMA void vstore(Fp *const dst[8], const V8 &a) {
    const V8 x = vmul(vnorm(a), vbroadcast(consts().from_v));             // x 2^384, below 2 p
    alignas(64) uint64_t st[8][8];
    for (int j = 0; j < 8; j++) _mm512_store_si512(st[j], x.l[j]);
    for (int lane = 0; lane < 8; lane++) {
        if (!dst[lane]) continue;
        unsigned __int128 acc = 0;
        int bits = 0, wi = 0;
        Fp r;
        for (int j = 0; j < 8; j++) {
            acc |= (unsigned __int128)st[j][lane] << bits;
            bits += 52;
            while (bits >= 32 && wi < 12) { r.l[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
        }
        while (wi < 12) { r.l[wi++] = (uint32_t)acc; acc >>= 32; }
        *dst[lane] = fp_reduce_once(r);
    }
}

// ---- cyclotomic squaring (Granger-Scott), state = (z0 .. z5) in lanes 0 .. 5 of two component vectors -------------------
// cyclotomic_sqr() of pairing.h:   (A0, A1) = sq4(z0, z1), (B0, B1) = sq4(z2, z3), (C0, C1) = sq4(z4, z5) with
//   sq4(a, b) = (a^2 + xi b^2, (a + b)^2 - a^2 - b^2);
//   z0' = 3 A0 - 2 z0,  z1' = 3 A1 + 2 z1,  z2' = 3 xi C1 + 2 z2,  z3' = 3 C0 - 2 z3,  z4' = 3 B0 - 2 z4,  z5' = 3 B1 + 2 z5.
struct Cyc { V8 c0, c1; };          // component 0 / 1 of z_lane; both normalised and below 3 p between squarings
EIP_IFMA Cyc cyc_sqr(const Cyc &z) {
    const __m512i swap = _mm512_setr_epi64(1, 0, 3, 2, 5, 4, 7, 6);
    // X = [z0 .. z5, z0 + z1, z2 + z3],  W = z4 + z5 (lane 4 of S)
    const V8 s0 = vadd(z.c0, vperm(z.c0, swap)), s1 = vadd(z.c1, vperm(z.c1, swap));
    const __m512i build = _mm512_setr_epi64(0, 1, 2, 3, 4, 5, 8 + 0, 8 + 2);
    const V8 x0 = vnorm(vperm2(z.c0, build, s0)), x1 = vnorm(vperm2(z.c1, build, s1));          // < 6 p
    // eight Fp2 squares: re = (x0 + x1)(x0 - x1), im / 2 = x0 x1
    const V8 re8 = vmul(vnorm(vadd(x0, x1)), vnorm(vsub64(x0, x1)));
    const V8 hf8 = vmul(x0, x1);
    // the ninth, (z4 + z5)^2, on lanes 0 (re) and 1 (im / 2) of a third product
    const V8 wp = vadd(s0, s1), wm = vsub64(s0, s1);
    const __m512i pick = _mm512_setr_epi64(4, 8 + 4, 4, 4, 4, 4, 4, 4);
    const V8 re9hf9 = vmul(vnorm(vperm2(wp, pick, s0)), vnorm(vperm2(wm, pick, s1)));          // lane 0: re, lane 1: im / 2
    const V8 im8 = vdbl(hf8);
    // squares by lane: 0..5 = z0^2 .. z5^2, 6 = (z0 + z1)^2, 7 = (z2 + z3)^2;  ninth: re = re9hf9[0], im = 2 re9hf9[1]
    // a^2 at even lanes, b^2 at odd lanes.  xi (b^2) = (re - im, re + im)
    const V8 xb_re = vsub64(re8, im8), xb_im = vadd(re8, im8);
    // T0 = a^2 + xi b^2 at the even lane of each pair: lanes 0 (A0), 2 (B0), 4 (C0)
    const __m512i odd_to_even = _mm512_setr_epi64(1, 1, 3, 3, 5, 5, 7, 7);
    const V8 t0_re = vadd(re8, vperm(xb_re, odd_to_even)), t0_im = vadd(im8, vperm(xb_im, odd_to_even));      // valid at lanes 0, 2, 4
    // T1 = s^2 - a^2 - b^2: A1 from lane 6, B1 from lane 7, C1 from the ninth
    const V8 ab_re = vadd(re8, vperm(re8, swap)), ab_im = vadd(im8, vperm(im8, swap));                            // a^2 + b^2 at lanes 0/1, 2/3, 4/5
    const __m512i nine_re = _mm512_setr_epi64(0, 0, 0, 0, 0, 0, 0, 0), nine_im = _mm512_setr_epi64(1, 1, 1, 1, 1, 1, 1, 1);
    const V8 s9_re = vperm(re9hf9, nine_re), s9_im = vdbl(vperm(re9hf9, nine_im));
    const __m512i s_src = _mm512_setr_epi64(6, 6, 7, 7, 8 + 0, 8 + 0, 6, 6);                                      // s^2 for pair A, B (from the eight) and C (ninth)
    const V8 t1_re = vsub64(vperm2(re8, s_src, s9_re), ab_re), t1_im = vsub64(vperm2(im8, s_src, s9_im), ab_im);  // valid at every lane of its pair
    // xi C1 for z2'
    const V8 xc_re = vsub512(t1_re, t1_im), xc_im = vadd(t1_re, t1_im);                                            // xi T1 (used at pair C's lanes only)
    // gather per output lane:  0: A0 (t0 lane 0)  1: A1 (t1 lane 1)  2: xi C1 (xc lane 4)  3: C0 (t0 lane 4)  4: B0 (t0 lane 2)  5: B1 (t1 lane 3)
    const __m512i g_t0 = _mm512_setr_epi64(0, 0, 0, 4, 2, 0, 0, 0);                    // lanes 0, 3, 4 taken from t0
    const __m512i g_t1 = _mm512_setr_epi64(1, 1, 8 + 4, 1, 1, 3, 1, 1);                // lanes 1, 5 from t1; lane 2 from xc (second source)
    const V8 u_re = vperm2(t1_re, g_t1, xc_re), u_im = vperm2(t1_im, g_t1, xc_im);
    const __mmask8 from_t0 = 0x19;                                                     // lanes 0, 3, 4
    const V8 T_re = vblend(from_t0, u_re, vperm(t0_re, g_t0)), T_im = vblend(from_t0, u_im, vperm(t0_im, g_t0));
    // z' = 3 T -+ 2 z:  minus at lanes 0, 3, 4;  plus at lanes 1, 2, 5;  lanes 6, 7 are cleared
    const V8 z2_re = vdbl(z.c0), z2_im = vdbl(z.c1);
    const V8 T3_re = vtriple(T_re), T3_im = vtriple(T_im);
    const V8 out_re = vblend(from_t0, vadd(T3_re, z2_re), vsub64(T3_re, z2_re));       // below ~2 000 p
    const V8 out_im = vblend(from_t0, vadd(T3_im, z2_im), vsub64(T3_im, z2_im));
    const __mmask8 live = 0x3f;
    Cyc r;
    r.c0 = vreduce(vnorm(vblend(live, vzero(), out_re)));
    r.c1 = vreduce(vnorm(vblend(live, vzero(), out_im)));
    return r;
}
EIP_IFMA Cyc cyc_load(const Fp12 &f) {
    // lanes z0 .. z5 = c0.a0, c1.a1, c1.a0, c0.a2, c0.a1, c1.a2   (the z numbering of cyclotomic_sqr())
    const Fp2 *z[6] = {&f.c0.a0, &f.c1.a1, &f.c1.a0, &f.c0.a2, &f.c0.a1, &f.c1.a2};
    const Fp *p0[8], *p1[8];
    for (int i = 0; i < 8; i++) { p0[i] = i < 6 ? &z[i]->c0 : nullptr; p1[i] = i < 6 ? &z[i]->c1 : nullptr; }
    return Cyc{vreduce(vload(p0)), vreduce(vload(p1))};
}
EIP_IFMA Fp12 cyc_store(const Cyc &c) {
    Fp12 f;
    Fp2 *z[6] = {&f.c0.a0, &f.c1.a1, &f.c1.a0, &f.c0.a2, &f.c0.a1, &f.c1.a2};
    Fp *p0[8], *p1[8];
    for (int i = 0; i < 8; i++) { p0[i] = i < 6 ? &z[i]->c0 : nullptr; p1[i] = i < 6 ? &z[i]->c1 : nullptr; }
    vstore(p0, c.c0);
    vstore(p1, c.c1);
    return f;
}
// ---- Fp12 products ---------------------------------------------------------------------------------------------------------
// F12v: lane k (k < 6) = the Fp2 coefficient of w^k  (w^6 = 1 + u; tower: c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)), as two
// component vectors, normalised, below 3 p.  A product is the schoolbook convolution in w, one shift s per round:
//     out_k += f_{k-s} g_s   (times 1 + u where k - s wraps),
// every round four accumulating 8-lane products (re: a0 b0 + (64 p - a1) b1,  im: a0 b1 + a1 b0) into two column sets that
// are reduced ONCE at the end -- 24 accumulations and 2 reductions per Fp12 product instead of 54 scalar products.
struct F12v { V8 c0, c1; };
EIP_IFMA F12v f12_mul(const F12v &f, const F12v &g) {
    const V8 zero = vzero();
    // forms of f:  f,  xi f = (f0 - f1, f0 + f1),  and the negated second components
    const V8 xf0 = vnorm(vsub64(f.c0, f.c1)), xf1 = vnorm(vadd(f.c0, f.c1));
    const V8 nf1 = vnorm(vsub64(zero, f.c1)), nxf1 = vnorm(vsub64(zero, xf1));
    __m512i tr[17], ti[17];
    for (int j = 0; j < 17; j++) { tr[j] = _mm512_setzero_si512(); ti[j] = _mm512_setzero_si512(); }
    for (int s = 0; s < 6; s++) {
        // lane k takes f_{k-s} (first source) or xi f_{k-s+6} (second source) where k < s
        long long ix[8];
        for (int k = 0; k < 8; k++) ix[k] = k >= 6 ? 6 : (k >= s ? k - s : 8 + (k - s + 6));
        const __m512i idx = _mm512_setr_epi64(ix[0], ix[1], ix[2], ix[3], ix[4], ix[5], ix[6], ix[7]);
        const V8 a0 = vperm2(f.c0, idx, xf0), a1 = vperm2(f.c1, idx, xf1), na1 = vperm2(nf1, idx, nxf1);
        const __m512i bs = _mm512_set1_epi64(s);
        const V8 b0 = vperm(g.c0, bs), b1 = vperm(g.c1, bs);
        vmac(tr, a0, b0);
        vmac(tr, na1, b1);
        vmac(ti, a0, b1);
        vmac(ti, a1, b0);
    }
    return F12v{vredc(tr), vredc(ti)};
}
EIP_IFMA F12v f12_conj(const F12v &a) {                       // c1 -> -c1: the odd powers of w
    const __mmask8 odd = 0x2a;
    const V8 zero = vzero();
    return F12v{vreduce(vnorm(vblend(odd, a.c0, vsub64(zero, a.c0)))), vreduce(vnorm(vblend(odd, a.c1, vsub64(zero, a.c1))))};   // below 3 p again
}
EIP_IFMA F12v f12_load(const Fp12 &f) {
    const Fp2 *w[6] = {&f.c0.a0, &f.c1.a0, &f.c0.a1, &f.c1.a1, &f.c0.a2, &f.c1.a2};
    const Fp *p0[8], *p1[8];
    for (int i = 0; i < 8; i++) { p0[i] = i < 6 ? &w[i]->c0 : nullptr; p1[i] = i < 6 ? &w[i]->c1 : nullptr; }
    return F12v{vload(p0), vload(p1)};
}
EIP_IFMA Fp12 f12_store(const F12v &v) {
    Fp12 f;
    Fp2 *w[6] = {&f.c0.a0, &f.c1.a0, &f.c0.a1, &f.c1.a1, &f.c0.a2, &f.c1.a2};
    Fp *p0[8], *p1[8];
    for (int i = 0; i < 8; i++) { p0[i] = i < 6 ? &w[i]->c0 : nullptr; p1[i] = i < 6 ? &w[i]->c1 : nullptr; }
    vstore(p0, v.c0);
    vstore(p1, v.c1);
    return f;
}
// w-order <-> the z-order of cyc_sqr():  z0 .. z5 = w^0, w^3, w^1, w^4, w^2, w^5
EIP_IFMA Cyc to_cyc(const F12v &a) {
    const __m512i idx = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7);
    return Cyc{vperm(a.c0, idx), vperm(a.c1, idx)};
}
EIP_IFMA F12v from_cyc(const Cyc &c) {
    const __m512i idx = _mm512_setr_epi64(0, 2, 4, 1, 3, 5, 6, 7);
    return F12v{vperm(c.c0, idx), vperm(c.c1, idx)};
}
// g^|z| for g in the cyclotomic subgroup, all of it on the vector unit
EIP_IFMA F12v f12_exp_zabs(const F12v &g) {
    const uint64_t z = K_Z_ABS;
    F12v acc = g;
    int i = 62;
    while (i >= 0) {
        Cyc c = to_cyc(acc);
        c.c0 = vreduce(c.c0);
        c.c1 = vreduce(c.c1);
        bool bit = false;
        do {                                        // square down to (and including) the next set bit, or the end
            c = cyc_sqr(c);
            bit = (z >> i) & 1ull;
            i--;
        } while (!bit && i >= 0);
        acc = from_cyc(c);
        if (bit) acc = f12_mul(acc, g);
    }
    return acc;
}
// final_exp() of pairing.h: the easy part (one inversion) in scalar code, the hard part on the vector unit
EIP_IFMA Fp12 final_exp_ifma(const Fp12 &f) {
    const Fp12 f1 = mul(conj(f), inv(f));
    const Fp12 f2s = mul(frob2(f1), f1);
    const F12v f2 = f12_load(f2s);
    auto ez = [](const F12v &g) { return f12_conj(f12_exp_zabs(g)); };          // z < 0
    const F12v y0 = f12_mul(ez(f2), f12_conj(f2));
    const F12v y1 = f12_mul(ez(y0), f12_conj(y0));
    const Fp12 y1s = f12_store(y1);
    const F12v y2 = f12_mul(ez(y1), f12_load(frob(y1s)));
    const Fp12 y2s = f12_store(y2);
    const F12v y3 = f12_mul(f12_mul(ez(ez(y2)), f12_load(frob2(y2s))), f12_conj(y2));
    return f12_store(f12_mul(y3, f12_mul(f12_mul(f2, f2), f2)));
}
// (...((L_0)^2 L_1)^2 ...) over the per-group products of a pairing batch (pairing.hip): `sq[g]` squarings, then times L_g
EIP_IFMA Fp12 horner_groups_ifma(const Fp12 *L, const int *sq, int n) {
    F12v F = f12_load(L[0]);
    for (int g = 1; g < n; g++) {
        for (int k = 0; k < sq[g]; k++) F = f12_mul(F, F);
        F = f12_mul(F, f12_load(L[g]));
    }
    return f12_store(F);
}

}  // namespace ifma
// The host's final exponentiation: the vector path where the CPU has AVX-512 IFMA, the scalar one otherwise
// (EIP2537_HOST_IFMA=0 forces the scalar path).
inline Fp12 final_exp_host(const Fp12 &f) {
    static const bool use = [] { const char *v = getenv("EIP2537_HOST_IFMA"); return ifma::cpu_has_ifma() && !(v && atoi(v) == 0); }();
    return use ? ifma::final_exp_ifma(f) : final_exp(f);
}
inline bool host_ifma_enabled() {
    static const bool use = [] { const char *v = getenv("EIP2537_HOST_IFMA"); return ifma::cpu_has_ifma() && !(v && atoi(v) == 0); }();
    return use;
}
}  // namespace eip
#else
namespace eip {
inline Fp12 final_exp_host(const Fp12 &f) { return final_exp(f); }
inline bool host_ifma_enabled() { return false; }
}
#endif
