// BLS12-381 base field Fp and Fp2 = Fp[u]/(u^2+1), Montgomery form (R = 2^384).
//
// One source for both sides of the engine: on the GPU (gfx950) an element is 12 x 32-bit
// limbs in VGPRs and the product is a CIOS loop of v_mad_u64_u32; on the host the same 48
// bytes are read as 6 x 64-bit limbs.  This layer replaces what the reference obtains from
// blst: blst_fp_add / blst_fp_to / blst_fp_from (reference src/eip2537.c:287,301; src/eip2537.h:15-16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "constants.h"

// HD: ordinary inline (the AMDGPU pipeline inlines every device function that is not marked
// noinline; the host compiler decides for itself).  The Fp product is the one deliberate
// out-of-line function on the device: a point addition is 10-14 products of ~900 instructions
// each, and inlining them all makes the code objects (and hipcc) explode.
#define HD __host__ __device__ inline

namespace eip {

struct Fp { uint32_t l[12]; };
struct Fp2 { Fp c0, c1; };

HD Fp fp_p() { return Fp{{K_P}}; }
HD Fp fp_zero() { return Fp{{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
HD Fp fp_one() { return Fp{{K_ONE}}; }

HD bool is_zero(const Fp &a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc |= a.l[i];
    return acc == 0;
}
HD bool eq(const Fp &a, const Fp &b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc |= a.l[i] ^ b.l[i];
    return acc == 0;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host paths: the same 48 bytes handled as 6 x 64-bit limbs.
namespace hostfp {
typedef unsigned long long ull;
struct L6 { uint64_t w[6]; };
inline L6 load(const Fp &a) { L6 r; memcpy(r.w, a.l, 48); return r; }
inline Fp store(const L6 &a) { Fp r; memcpy(r.l, a.w, 48); return r; }
// K_P's 32-bit words paired into 64-bit limbs at compile time
constexpr uint32_t kP32[12] = {K_P};
constexpr uint64_t p64(int i) { return (uint64_t)kP32[2 * i] | ((uint64_t)kP32[2 * i + 1] << 32); }
constexpr uint64_t kP64[6] = {p64(0), p64(1), p64(2), p64(3), p64(4), p64(5)};
inline L6 modulus() { return L6{{kP64[0], kP64[1], kP64[2], kP64[3], kP64[4], kP64[5]}}; }
inline L6 reduce_once(const L6 &t) {
    ull d[6], borrow = 0;
    for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(t.w[i], kP64[i], borrow, &borrow);
    L6 r;
    for (int i = 0; i < 6; i++) r.w[i] = borrow ? t.w[i] : d[i];
    return r;
}
inline L6 add(const L6 &a, const L6 &b) {
    L6 t;
    ull c = 0;
    for (int i = 0; i < 6; i++) t.w[i] = __builtin_addcll(a.w[i], b.w[i], c, &c);
    return reduce_once(t);          // a + b < 2p < 2^384: no carry out
}
inline L6 sub(const L6 &a, const L6 &b) {
    ull d[6], borrow = 0, c = 0;
    for (int i = 0; i < 6; i++) d[i] = __builtin_subcll(a.w[i], b.w[i], borrow, &borrow);
    const uint64_t mask = 0 - (uint64_t)borrow;
    L6 r;
    for (int i = 0; i < 6; i++) r.w[i] = __builtin_addcll(d[i], kP64[i] & mask, c, &c);
    return r;
}
}  // namespace hostfp
#endif

// r = t - p if t >= p else t, for t < 2p (t fits in 384 bits)
HD Fp fp_reduce_once(const Fp &t) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return hostfp::store(hostfp::reduce_once(hostfp::load(t)));
#else
    // device: native carry chains (v_subb_co_u32); the 64-bit-arithmetic spelling of the same loop
    // compiled to v_lshl_add_u64 + zero-extension moves, ~3x the instructions
    const Fp p = fp_p();
    Fp d;
    unsigned borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = __builtin_subc(t.l[i], p.l[i], borrow, &borrow);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = borrow ? t.l[i] : d.l[i];
    return r;
#endif
}
HD Fp add(const Fp &a, const Fp &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return hostfp::store(hostfp::add(hostfp::load(a), hostfp::load(b)));
#else
    Fp t;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) t.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    return fp_reduce_once(t);     // a + b < 2p < 2^384: no carry out
#endif
}
HD Fp sub(const Fp &a, const Fp &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return hostfp::store(hostfp::sub(hostfp::load(a), hostfp::load(b)));
#else
    const Fp p = fp_p();
    Fp d, e;
    unsigned borrow = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = __builtin_subc(a.l[i], b.l[i], borrow, &borrow);
#pragma unroll
    for (int i = 0; i < 12; i++) e.l[i] = __builtin_addc(d.l[i], p.l[i], c, &c);
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = borrow ? e.l[i] : d.l[i];
    return r;
#endif
}
HD Fp neg(const Fp &a) { return sub(fp_zero(), a); }
HD Fp dbl(const Fp &a) { return add(a, a); }

// Montgomery product, 12 x 32-bit CIOS (round-1 first version; kept as the comparison point of
// tools/fpmul_bench.hip and as a host-side cross-check of fp_mul_cols28).
HD Fp fp_mul_limbs32(const Fp &a, const Fp &b) {
    const Fp p = fp_p();
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 13; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint64_t s = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        t[12] = c;
        const uint32_t m = t[0] * K_N0_32;
        uint64_t s = (uint64_t)m * p.l[0] + t[0];
        c = (uint32_t)(s >> 32);
#pragma unroll
        for (int j = 1; j < 12; j++) {
            s = (uint64_t)m * p.l[j] + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        t[11] = t[12] + c;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = t[i];
    return fp_reduce_once(r);
}

// ---- radix 2^30 (13 limbs) ------------------------------------------------------------------------------
// 13 x 13 + 13 x 13 = 338 multiply-adds instead of the 392 of the 14 x 28-bit form below.  A column of
// 26 products of 30-bit limbs would overflow its 64-bit accumulator by 0.7 bit; with the real limbs of p the
// reduction's share of a column is at most 6.9 x 2^60, and carrying columns 10 .. 14 out ONCE, after outer step 7,
// keeps every column below 2^64 (exact worst case per column: tools/limb_column_bounds.py; round 2 carried
// 8 .. 16 with a full shift / add / mask each).  The
// reduction clears 12 x 30 + 24 = 384 bits, so R = 2^384 and the results are bit-identical to every other
// product in this file.  Inputs may be in [0, 2p) (FpI); REDUCE = false leaves the result below 1.41 p.
// hipcc (ROCm 7.2, gfx950) miscompiles the 13-limb product when it can see that the last reduction
// factor is "x & 0xffffff": the 24-bit multiply combine drops the mask (its instruction ignores bits
// 24..31), and the sum of the two 24 x 24-bit products of column 24 is then selected as v_mad_u64_u32
// on the UNMASKED register (tools/dev_mul_check.hip: 99.6 % wrong products on the device, none on the
// host). The 14 x 28-bit product masks its last factor to 20 bits and is not affected. The empty asm
// hides a value's range from the optimiser and emits no instruction; it is applied to the reduction
// factors and to the limbs.
#if defined(__HIP_DEVICE_COMPILE__)
#define EIP_OPAQUE(x) asm("" : "+v"(x))
#else
#define EIP_OPAQUE(x)
#endif
// Carry-out of a 64-bit column accumulator of the 30-bit-limb products: the HIGH DWORD moves into the next column (x 4:
// 2^32 = 4 x 2^30), the column keeps its low 32 bits.  Which columns need it and when: limb30.h / tools/limb_column_bounds.py.
HD void col_carry_hi(uint64_t *col, int c) {
    const uint32_t hi = (uint32_t)(col[c] >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    // (the compiler turns hi * 4 into a 64-bit shift, two masks and a 64-bit add)
    asm("v_mad_u64_u32 %0, vcc, %1, 4, %2" : "=v"(col[c + 1]) : "v"(hi), "v"(col[c + 1]) : "vcc");
#else
    col[c + 1] += (uint64_t)hi << 2;
#endif
    col[c] &= 0xffffffffull;
}
template <bool REDUCE> HD Fp fp_mul_cols30_t(const Fp &a, const Fp &b) {
    const uint32_t p30[13] = {K_P30};
    const uint32_t M30 = 0x3fffffffu;
    uint32_t al[13], bl[13];
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int bit = 30 * k, i = bit >> 5, s = bit & 31;
        uint32_t va = a.l[i] >> s, vb = b.l[i] >> s;
        if (s > 2 && i + 1 < 12) { va |= a.l[i + 1] << (32 - s); vb |= b.l[i + 1] << (32 - s); }
        al[k] = va & M30;
        bl[k] = vb & M30;
        EIP_OPAQUE(al[k]);
        EIP_OPAQUE(bl[k]);
    }
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)al[j] * bl[i];
        // last step clears only 24 bits: 12 * 30 + 24 = 384
        uint32_t m = ((uint32_t)col[i] * K_N0_30) & (i < 12 ? M30 : 0x00ffffffu);
        EIP_OPAQUE(m);
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        if (i < 12) col[i + 1] += col[i] >> 30;
        if (i == 7) {
#pragma unroll
            for (int c = 10; c <= 14; c++) col_carry_hi(col, c);
        }
    }
    // digits 12..25 hold (result << 24); propagate carries, then cut 32-bit words at bit 24
    uint32_t d[16];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint64_t v = col[12 + k] + carry;
        d[k] = (uint32_t)v & M30;
        carry = v >> 30;
    }
    d[14] = 0;
    d[15] = 0;
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 24 + 32 * w, q = bit / 30, o = bit % 30;
        uint64_t t = (uint64_t)d[q] | ((uint64_t)d[q + 1] << 30) | ((uint64_t)d[q + 2] << 60);
        r.l[w] = (uint32_t)(t >> o);
    }
    return REDUCE ? fp_reduce_once(r) : r;
}
// Squaring: the 78 cross products are computed once against a doubled operand (91 + 169 multiply-adds);
// the product phase leaves at most 13 terms per column, columns 10 .. 14 are carried out before the
// reduction adds its terms (limb30.h).
template <bool REDUCE> HD Fp fp_sqr_cols30_t(const Fp &a) {
    const uint32_t p30[13] = {K_P30};
    const uint32_t M30 = 0x3fffffffu;
    uint32_t al[13], a2[13];
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int bit = 30 * k, i = bit >> 5, s = bit & 31;
        uint32_t va = a.l[i] >> s;
        if (s > 2 && i + 1 < 12) va |= a.l[i + 1] << (32 - s);
        al[k] = va & M30;
        EIP_OPAQUE(al[k]);
        a2[k] = al[k] << 1;
        EIP_OPAQUE(a2[k]);
    }
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        col[2 * i] += (uint64_t)al[i] * al[i];
#pragma unroll
        for (int j = i + 1; j < 13; j++) col[i + j] += (uint64_t)a2[j] * al[i];
    }
#pragma unroll
    for (int c = 10; c <= 14; c++) col_carry_hi(col, c);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        uint32_t m = ((uint32_t)col[i] * K_N0_30) & (i < 12 ? M30 : 0x00ffffffu);
        EIP_OPAQUE(m);
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        if (i < 12) col[i + 1] += col[i] >> 30;
    }
    uint32_t d[16];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint64_t v = col[12 + k] + carry;
        d[k] = (uint32_t)v & M30;
        carry = v >> 30;
    }
    d[14] = 0;
    d[15] = 0;
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 24 + 32 * w, q = bit / 30, o = bit % 30;
        uint64_t t = (uint64_t)d[q] | ((uint64_t)d[q + 1] << 30) | ((uint64_t)d[q + 2] << 60);
        r.l[w] = (uint32_t)(t >> o);
    }
    return REDUCE ? fp_reduce_once(r) : r;
}

// (a b + c d) / 2^384 with ONE reduction: the component of an Fp2 product (u0 v0 - u1 v1, u0 v1 + u1 v0)
// in 507 multiply-adds instead of two products and an addition (676 + a carry chain).  Inputs in [0, 2p):
// the sum is below 8 p^2, so the result is below 8 p^2 / R + p = 1.82 p -- inside the lazy range of FpI.
// The rows of the two products alternate, columns 6 .. 18 are carried out after row 7 of both and column 12 before the
// reduction (limb30.h); the reduction is the one of fp_mul_cols30_t (its last factor hidden from the optimiser, see above).
HD Fp fp_mul2_cols30(const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
    const uint32_t p30[13] = {K_P30};
    const uint32_t M30 = 0x3fffffffu;
    uint32_t al[13], bl[13], cl[13], dl[13];
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int bit = 30 * k, i = bit >> 5, s = bit & 31;
        uint32_t va = a.l[i] >> s, vb = b.l[i] >> s, vc = c.l[i] >> s, vd = d.l[i] >> s;
        if (s > 2 && i + 1 < 12) {
            va |= a.l[i + 1] << (32 - s); vb |= b.l[i + 1] << (32 - s);
            vc |= c.l[i + 1] << (32 - s); vd |= d.l[i + 1] << (32 - s);
        }
        al[k] = va & M30; bl[k] = vb & M30; cl[k] = vc & M30; dl[k] = vd & M30;
        EIP_OPAQUE(al[k]); EIP_OPAQUE(bl[k]); EIP_OPAQUE(cl[k]); EIP_OPAQUE(dl[k]);
    }
    uint64_t col[27];
#pragma unroll
    for (int i = 0; i < 27; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)al[j] * bl[i];
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)cl[j] * dl[i];
        if (i == 7) {
#pragma unroll
            for (int k = 6; k <= 18; k++) col_carry_hi(col, k);
        }
    }
    col_carry_hi(col, 12);
#pragma unroll
    for (int i = 0; i < 13; i++) {
        uint32_t m = ((uint32_t)col[i] * K_N0_30) & (i < 12 ? M30 : 0x00ffffffu);
        EIP_OPAQUE(m);
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        if (i < 12) col[i + 1] += col[i] >> 30;
    }
    uint32_t dg[16];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint64_t v = col[12 + k] + carry;
        dg[k] = (uint32_t)v & M30;
        carry = v >> 30;
    }
    dg[14] = 0;
    dg[15] = 0;
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 24 + 32 * w, q = bit / 30, o = bit % 30;
        uint64_t t = (uint64_t)dg[q] | ((uint64_t)dg[q + 1] << 30) | ((uint64_t)dg[q + 2] << 60);
        r.l[w] = (uint32_t)(t >> o);
    }
    return r;
}

// Device Montgomery product.  The operands arrive as canonical 12 x 32-bit limbs and are
// re-sliced into 14 x 28-bit limbs so that every multiply-add of the schoolbook product and of
// the reduction is ONE v_mad_u64_u32 accumulating in place into a 64-bit column: 28 terms of
// < 2^56 never overflow, so no carry is handled inside the loops (the 32-bit-limb CIOS spent
// more than half of its instructions on carry plumbing; measured 1.9x slower, profiles/).
// Reduction removes 13 x 28 + 20 = 384 bits, so R stays 2^384 and the result is bit-identical
// to the host's 6 x 64-bit product.
template <bool REDUCE> HD Fp fp_mul_cols28_t(const Fp &a, const Fp &b) {
    const uint32_t p28[14] = {K_P28};
    const uint32_t M28 = 0x0fffffffu;
    uint32_t al[14], bl[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const int bit = 28 * k, i = bit >> 5, s = bit & 31;
        uint32_t va = a.l[i] >> s, vb = b.l[i] >> s;
        if (s > 4 && i + 1 < 12) { va |= a.l[i + 1] << (32 - s); vb |= b.l[i + 1] << (32 - s); }
        al[k] = va & M28;
        bl[k] = vb & M28;
    }
    uint64_t col[28];
#pragma unroll
    for (int i = 0; i < 28; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
#pragma unroll
        for (int j = 0; j < 14; j++) col[i + j] += (uint64_t)al[j] * bl[i];
        // last step clears only 20 bits: 13 * 28 + 20 = 384
        const uint32_t m = ((uint32_t)col[i] * K_N0_28) & (i < 13 ? M28 : 0x000fffffu);
#pragma unroll
        for (int j = 0; j < 14; j++) col[i + j] += (uint64_t)m * p28[j];
        if (i < 13) col[i + 1] += col[i] >> 28;
    }
    // digits 13..27 hold (result << 20); propagate carries, then cut 32-bit words at bit 20
    uint32_t d[16];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        uint64_t v = col[13 + k] + carry;
        d[k] = (uint32_t)v & M28;
        carry = v >> 28;
    }
    d[15] = 0;
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 20 + 32 * w, q = bit / 28, o = bit % 28;
        uint64_t t = (uint64_t)d[q] | ((uint64_t)d[q + 1] << 28) | ((uint64_t)(q + 2 < 16 ? d[q + 2] : 0u) << 56);
        r.l[w] = (uint32_t)(t >> o);
    }
    return REDUCE ? fp_reduce_once(r) : r;
}
// Canonical result in [0, p).
HD Fp fp_mul_cols28(const Fp &a, const Fp &b) { return fp_mul_cols28_t<true>(a, b); }

// Squaring: the 91 cross products are computed once against a doubled operand.
template <bool REDUCE> HD Fp fp_sqr_cols28_t(const Fp &a) {
    const uint32_t p28[14] = {K_P28};
    const uint32_t M28 = 0x0fffffffu;
    uint32_t al[14], a2[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        const int bit = 28 * k, i = bit >> 5, s = bit & 31;
        uint32_t va = a.l[i] >> s;
        if (s > 4 && i + 1 < 12) va |= a.l[i + 1] << (32 - s);
        al[k] = va & M28;
        a2[k] = al[k] << 1;
    }
    uint64_t col[28];
#pragma unroll
    for (int i = 0; i < 28; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        col[2 * i] += (uint64_t)al[i] * al[i];
#pragma unroll
        for (int j = i + 1; j < 14; j++) col[i + j] += (uint64_t)a2[j] * al[i];
    }
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t m = ((uint32_t)col[i] * K_N0_28) & (i < 13 ? M28 : 0x000fffffu);
#pragma unroll
        for (int j = 0; j < 14; j++) col[i + j] += (uint64_t)m * p28[j];
        if (i < 13) col[i + 1] += col[i] >> 28;
    }
    uint32_t d[16];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        uint64_t v = col[13 + k] + carry;
        d[k] = (uint32_t)v & M28;
        carry = v >> 28;
    }
    d[15] = 0;
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 20 + 32 * w, q = bit / 28, o = bit % 28;
        uint64_t t = (uint64_t)d[q] | ((uint64_t)d[q + 1] << 28) | ((uint64_t)(q + 2 < 16 ? d[q + 2] : 0u) << 56);
        r.l[w] = (uint32_t)(t >> o);
    }
    return REDUCE ? fp_reduce_once(r) : r;
}
HD Fp fp_sqr_cols28(const Fp &a) { return fp_sqr_cols28_t<true>(a); }

// The product the device kernels use: radix 2^30 (13 limbs, 338 multiply-adds) by default, measured
// 3 % faster end to end than radix 2^28 (14 limbs, 392) on all three workloads;
// -DEIP_LIMB_BITS=28 builds the library on the 28-bit form.
#ifndef EIP_LIMB_BITS
#define EIP_LIMB_BITS 30
#endif
template <bool REDUCE> HD Fp fp_mul_cols_t(const Fp &a, const Fp &b) {
#if EIP_LIMB_BITS == 30
    return fp_mul_cols30_t<REDUCE>(a, b);
#else
    return fp_mul_cols28_t<REDUCE>(a, b);
#endif
}
template <bool REDUCE> HD Fp fp_sqr_cols_t(const Fp &a) {
#if EIP_LIMB_BITS == 30
    return fp_sqr_cols30_t<REDUCE>(a);
#else
    return fp_sqr_cols28_t<REDUCE>(a);
#endif
}
// Canonical results in [0, p).
HD Fp fp_mul_cols(const Fp &a, const Fp &b) { return fp_mul_cols_t<true>(a, b); }
HD Fp fp_sqr_cols(const Fp &a) { return fp_sqr_cols_t<true>(a); }


#if !defined(__HIP_DEVICE_COMPILE__)
// Host: the same 48 bytes as 6 x 64-bit limbs.  Operand scanning with whole rows of 64x64->128
// products (mulx with -mbmi2) and explicit add-with-carry chains; the running value stays below
// 2p, so the seventh limb never overflows.  (The obvious "mac with a 128-bit accumulator" loop
// compiled to ~2x the cycles: every step waited on the previous carry.)
inline uint64_t host_mul_wide(uint64_t a, uint64_t b, unsigned long long *hi) {
    unsigned __int128 p = (unsigned __int128)a * b;
    *hi = (unsigned long long)(p >> 64);
    return (uint64_t)p;
}
inline Fp fp_mul_limbs64(const Fp &a, const Fp &b) {
    typedef unsigned long long ull;
    const hostfp::L6 la = hostfp::load(a), lb = hostfp::load(b);
    const uint64_t *A = la.w, *B = lb.w, *Pm = hostfp::kP64;
    ull t[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ull lo[6], hi[6], c = 0;
        for (int j = 0; j < 6; j++) lo[j] = host_mul_wide(A[j], B[i], &hi[j]);
        for (int j = 0; j < 6; j++) t[j] = __builtin_addcll(t[j], lo[j], c, &c);
        t[6] = __builtin_addcll(t[6], 0, c, &c);
        c = 0;
        for (int j = 0; j < 6; j++) t[j + 1] = __builtin_addcll(t[j + 1], hi[j], c, &c);
        const ull m = t[0] * K_N0_64;
        for (int j = 0; j < 6; j++) lo[j] = host_mul_wide(m, Pm[j], &hi[j]);
        c = 0;
        (void)__builtin_addcll(t[0], lo[0], c, &c);           // low limb cancels by construction
        for (int j = 1; j < 6; j++) t[j - 1] = __builtin_addcll(t[j], lo[j], c, &c);
        t[5] = __builtin_addcll(t[6], 0, c, &c);
        t[6] = 0;
        c = 0;
        for (int j = 0; j < 6; j++) t[j] = __builtin_addcll(t[j], hi[j], c, &c);
    }
    hostfp::L6 r;
    for (int i = 0; i < 6; i++) r.w[i] = t[i];
    return hostfp::store(hostfp::reduce_once(r));
}
#endif

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__) && defined(__ADX__) && defined(__BMI2__) && !defined(EIP_HOST_NO_ASM)
#define EIP_HOST_ADX 1
// Host Montgomery product with the two independent carry chains of ADX (adcx: low halves, adox: high
// halves) over mulx rows -- the same CIOS schedule as fp_mul_limbs64 above, which compilers serialise
// on one carry flag (51 ns per product on the GPU box's host against ~30 ns here).  Per outer step:
// t += a * b_i, m = t0 * n0, t += m * p, and the limb names rotate by one (t0 has become zero and
// serves as the next step's top limb), so there is no shifting at all.  Everything stays below 2p + 1
// ulp, hence in 6 limbs + the rotating zero; the final conditional subtraction is hostfp::reduce_once.
// Host-side products are the serial tails of the engine: the window Horner of an MSM, the 63-squaring
// Horner and the final exponentiation of a pairing check, the single-pair precompiles and the
// small-call route (api.hip).
#define EIP_ADX_ROW(src, T0, T1, T2, T3, T4, T5, T6)                                             \
    "xorl %k[z], %k[z]\n\t"                                                                      \
    "mulxq 0(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T0 "]\n\t adoxq %[hi], %[" T1 "]\n\t"  \
    "mulxq 8(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T1 "]\n\t adoxq %[hi], %[" T2 "]\n\t"  \
    "mulxq 16(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T2 "]\n\t adoxq %[hi], %[" T3 "]\n\t" \
    "mulxq 24(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T3 "]\n\t adoxq %[hi], %[" T4 "]\n\t" \
    "mulxq 32(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T4 "]\n\t adoxq %[hi], %[" T5 "]\n\t" \
    "mulxq 40(%[" src "]), %[lo], %[hi]\n\t adcxq %[lo], %[" T5 "]\n\t adoxq %[hi], %[" T6 "]\n\t" \
    "adcxq %[z], %[" T6 "]\n\t"
#define EIP_ADX_STEP(off, T0, T1, T2, T3, T4, T5, T6)                                            \
    "movq " off "(%[b]), %%rdx\n\t" EIP_ADX_ROW("a", T0, T1, T2, T3, T4, T5, T6)                   \
    "movq %[" T0 "], %%rdx\n\t imulq %[n0], %%rdx\n\t" EIP_ADX_ROW("p", T0, T1, T2, T3, T4, T5, T6)
inline Fp fp_mul_adx(const Fp &a, const Fp &b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, lo, hi, z;
    const uint64_t n0 = K_N0_64;
    __asm__(EIP_ADX_STEP("0", "t0", "t1", "t2", "t3", "t4", "t5", "t6")
            EIP_ADX_STEP("8", "t1", "t2", "t3", "t4", "t5", "t6", "t0")
            EIP_ADX_STEP("16", "t2", "t3", "t4", "t5", "t6", "t0", "t1")
            EIP_ADX_STEP("24", "t3", "t4", "t5", "t6", "t0", "t1", "t2")
            EIP_ADX_STEP("32", "t4", "t5", "t6", "t0", "t1", "t2", "t3")
            EIP_ADX_STEP("40", "t5", "t6", "t0", "t1", "t2", "t3", "t4")
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [t5] "+&r"(t5), [t6] "+&r"(t6),
              [lo] "=&r"(lo), [hi] "=&r"(hi), [z] "=&r"(z)
            : [a] "r"(a.l), [b] "r"(b.l), [p] "r"(hostfp::kP64), [n0] "rm"(n0)
            : "rdx", "cc", "memory");
    // after six rotations the value sits in t6, t0, t1, t2, t3, t4 (t5 is the spent zero)
    hostfp::L6 r{{t6, t0, t1, t2, t3, t4}};
    return hostfp::store(hostfp::reduce_once(r));
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#if defined(EIP_DEBUG_OLD_MUL)
static __device__ __noinline__ Fp fp_mul_outlined(Fp a, Fp b) { return fp_mul_limbs32(a, b); }
static __device__ __noinline__ Fp fp_sqr_outlined(Fp a) { return fp_mul_limbs32(a, a); }
#else
static __device__ __noinline__ Fp fp_mul_outlined(Fp a, Fp b) { return fp_mul_cols(a, b); }
static __device__ __noinline__ Fp fp_sqr_outlined(Fp a) { return fp_sqr_cols(a); }
#endif
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
inline Fp fp_mul_host(const Fp &a, const Fp &b) {
#if defined(EIP_HOST_ADX)
    return fp_mul_adx(a, b);
#else
    return fp_mul_limbs64(a, b);
#endif
}
#endif
HD Fp mul(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_mul_outlined(a, b);
#else
    return fp_mul_host(a, b);
#endif
}
HD Fp sqr(const Fp &a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_sqr_outlined(a);
#else
    return fp_mul_host(a, a);
#endif
}

// a^e for a plain little-endian exponent of nwords 32-bit words (MSB-first square and multiply)
HD Fp fp_pow(const Fp &a, const uint32_t *e, int nwords) {
    Fp acc = fp_one();
    for (int i = nwords * 32 - 1; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = mul(acc, a);
    }
    return acc;
}
// Fermat inverse; inv(0) = 0 so that to-affine of infinity is (0,0) (what the reference
// relies on from blst_p1_to_affine).
HD Fp inv(const Fp &a) {
    const uint32_t e[12] = {K_P_MINUS_2};
    return fp_pow(a, e, 12);
}
HD Fp fp_to_mont(const Fp &raw) { return mul(raw, Fp{{K_RR}}); }
HD Fp fp_from_mont(const Fp &a) { return mul(a, Fp{{1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}); }

// ---------------------------------------------------------------------------- Fp2
HD Fp2 fp2_zero() { return Fp2{fp_zero(), fp_zero()}; }
HD Fp2 fp2_one() { return Fp2{fp_one(), fp_zero()}; }
HD bool is_zero(const Fp2 &a) { return is_zero(a.c0) && is_zero(a.c1); }
HD bool eq(const Fp2 &a, const Fp2 &b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
HD Fp2 add(const Fp2 &a, const Fp2 &b) { return Fp2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
HD Fp2 sub(const Fp2 &a, const Fp2 &b) { return Fp2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
HD Fp2 neg(const Fp2 &a) { return Fp2{neg(a.c0), neg(a.c1)}; }
HD Fp2 dbl(const Fp2 &a) { return Fp2{dbl(a.c0), dbl(a.c1)}; }
HD Fp2 conj(const Fp2 &a) { return Fp2{a.c0, neg(a.c1)}; }
// Fp products as seen from inside the Fp2 bodies: inlined on the device (the Fp2 product itself
// is the out-of-line unit there, so a nested call per Fp product would only add callee-save
// spill traffic), the 64-bit host product otherwise.
HD Fp fp_mul_leaf(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_mul_cols(a, b);
#else
    return fp_mul_host(a, b);
#endif
}
HD Fp2 fp2_mul_body(const Fp2 &a, const Fp2 &b) {
#if defined(__HIP_DEVICE_COMPILE__) && EIP_LIMB_BITS == 30
    // schoolbook with one reduction per component (fp_mul2_cols30): the same 1 014 multiply-adds as the three
    // Karatsuba products below without their two additions and three subtractions.  Canonical operands:
    // each sum is below 2 p^2, the result below 1.21 p.
    return Fp2{fp_reduce_once(fp_mul2_cols30(a.c0, b.c0, a.c1, neg(b.c1))),
               fp_reduce_once(fp_mul2_cols30(a.c0, b.c1, a.c1, b.c0))};
#endif
    Fp t0 = fp_mul_leaf(a.c0, b.c0);
    Fp t1 = fp_mul_leaf(a.c1, b.c1);
    Fp t2 = fp_mul_leaf(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fp2{sub(t0, t1), sub(sub(t2, t0), t1)};
}
HD Fp2 fp2_sqr_body(const Fp2 &a) {
    Fp m = fp_mul_leaf(a.c0, a.c1);
    return Fp2{fp_mul_leaf(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(m)};
}
// On the device the Fp2 product and square are out of line as well.  Besides keeping the code
// objects small this is a correctness guard: with fp_mul_cols and everything above it inlined,
// hipcc 7.2 -O3 miscompiled k_pair_check_g2 (256 VGPR + 256 AGPR + scratch spills; the same
// source was correct at -O1, with the 32-bit CIOS product, and in a smaller kernel) -- found by
// the GPU parity tests, isolated by compiling that kernel alone at -O1 / -O3 and with either product.
#if defined(__HIP_DEVICE_COMPILE__)
// Fp2 product as three calls of the out-of-line Fp product.  Two Fp operands (24 dwords) travel in
// argument VGPRs; two Fp2 operands (48 dwords) do not fit the 32 argument registers and go through
// the stack, which made every fp2_mul_outlined call ~12 KB of scratch traffic per wave
// (profiles/r01_pmc_pairing_2p12.csv: k_pair_tree wrote 2 GB per launch that way).
static __device__ __forceinline__ Fp2 fp2_mul_regcall(const Fp2 &a, const Fp2 &b) {
    Fp t0 = fp_mul_outlined(a.c0, b.c0);
    Fp t1 = fp_mul_outlined(a.c1, b.c1);
    Fp t2 = fp_mul_outlined(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fp2{sub(t0, t1), sub(sub(t2, t0), t1)};
}
static __device__ __forceinline__ Fp2 fp2_sqr_regcall(const Fp2 &a) {
    Fp m = fp_mul_outlined(a.c0, a.c1);
    return Fp2{fp_mul_outlined(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(m)};
}
static __device__ __noinline__ Fp2 fp2_mul_outlined(Fp2 a, Fp2 b) { return fp2_mul_body(a, b); }
static __device__ __noinline__ Fp2 fp2_sqr_outlined(Fp2 a) { return fp2_sqr_body(a); }
#ifndef EIP_FP2_POLICY
#define EIP_FP2_POLICY 1
#endif
#if EIP_FP2_POLICY == 1
HD Fp2 mul(const Fp2 &a, const Fp2 &b) { return fp2_mul_regcall(a, b); }
HD Fp2 sqr(const Fp2 &a) { return fp2_sqr_regcall(a); }
#else
HD Fp2 mul(const Fp2 &a, const Fp2 &b) { return fp2_mul_outlined(a, b); }
HD Fp2 sqr(const Fp2 &a) { return fp2_sqr_outlined(a); }
#endif
#else
HD Fp2 mul(const Fp2 &a, const Fp2 &b) { return fp2_mul_body(a, b); }
HD Fp2 sqr(const Fp2 &a) { return fp2_sqr_body(a); }
#endif
HD Fp2 mul_fp(const Fp2 &a, const Fp &s) { return Fp2{mul(a.c0, s), mul(a.c1, s)}; }
HD Fp2 mul_xi(const Fp2 &a) { return Fp2{sub(a.c0, a.c1), add(a.c0, a.c1)}; }   // * (1 + u)
HD Fp2 inv(const Fp2 &a) {
    Fp n = inv(add(sqr(a.c0), sqr(a.c1)));
    return Fp2{mul(a.c0, n), neg(mul(a.c1, n))};
}


// FpI: the same 48 bytes as Fp, with two differences that matter only to the kernels that are bound by
// the instruction count of their field arithmetic (k_msm_accum<Fp>, the G1 fold / reduce kernels):
//  * its product and square are inlined at every use (the call / argument-shuffle overhead of the
//    out-of-line product is measurable when one mixed addition is the whole loop body), and
//  * on the device its values are only kept in [0, 2p), not [0, p): the Montgomery product of two such
//    values is below (4p^2 + R p) / R < 1.41 p (R = 2^384 > 9.8 p), so the product needs NO final
//    conditional subtraction (36 instructions + 24 hazard nops of ~620), and addition / subtraction
//    wrap at 2p for the same price as they wrapped at p.  Zero is {0, p}.  Canonical Fp values are valid
//    FpI values (decoded points go in as they are); values leave through fp_canon().
struct FpI { Fp v; };
HD Fp fp_2p() {
    const Fp p = fp_p();
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = (p.l[i] << 1) | (i ? p.l[i - 1] >> 31 : 0u);
    return r;
}
HD Fp fp_canon(const FpI &a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_reduce_once(a.v);
#else
    return a.v;
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
HD bool is_zero(const FpI &a) {
    const Fp p = fp_p();
    uint32_t z = 0, zp = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { z |= a.v.l[i]; zp |= a.v.l[i] ^ p.l[i]; }
    return z == 0 || zp == 0;
}
HD FpI add(const FpI &a, const FpI &b) {
    const Fp m = fp_2p();
    Fp t, d;
    unsigned c = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) t.l[i] = __builtin_addc(a.v.l[i], b.v.l[i], c, &c);          // < 4p < 2^384
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = __builtin_subc(t.l[i], m.l[i], borrow, &borrow);
    FpI r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.v.l[i] = borrow ? t.l[i] : d.l[i];
    return r;
}
HD FpI sub(const FpI &a, const FpI &b) {
    const Fp m = fp_2p();
    Fp d, e;
    unsigned borrow = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d.l[i] = __builtin_subc(a.v.l[i], b.v.l[i], borrow, &borrow);
#pragma unroll
    for (int i = 0; i < 12; i++) e.l[i] = __builtin_addc(d.l[i], m.l[i], c, &c);
    FpI r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.v.l[i] = borrow ? e.l[i] : d.l[i];
    return r;
}
HD FpI mul(const FpI &a, const FpI &b) { return FpI{fp_mul_cols_t<false>(a.v, b.v)}; }
HD FpI sqr(const FpI &a) { return FpI{fp_sqr_cols_t<false>(a.v)}; }
// a b + c d, reduced once (the radix-2^30 build; two products and an addition otherwise)
HD FpI mul2(const FpI &a, const FpI &b, const FpI &c, const FpI &d) {
#if EIP_LIMB_BITS == 30
    return FpI{fp_mul2_cols30(a.v, b.v, c.v, d.v)};
#else
    return add(mul(a, b), mul(c, d));
#endif
}
#else       // the host pass only parses the kernels that use FpI
HD bool is_zero(const FpI &a) { return is_zero(a.v); }
HD FpI add(const FpI &a, const FpI &b) { return FpI{add(a.v, b.v)}; }
HD FpI sub(const FpI &a, const FpI &b) { return FpI{sub(a.v, b.v)}; }
HD FpI mul(const FpI &a, const FpI &b) { return FpI{fp_mul_host(a.v, b.v)}; }
HD FpI sqr(const FpI &a) { return FpI{fp_mul_host(a.v, a.v)}; }
HD FpI mul2(const FpI &a, const FpI &b, const FpI &c, const FpI &d) { return FpI{add(fp_mul_host(a.v, b.v), fp_mul_host(c.v, d.v))}; }
#endif
HD bool eq(const FpI &a, const FpI &b) { return is_zero(sub(a, b)); }
HD FpI neg(const FpI &a) { return sub(FpI{fp_zero()}, a); }
HD FpI dbl(const FpI &a) { return add(a, a); }

// uniform spelling for the curve templates
template <class F> HD F f_zero();
template <class F> HD F f_one();
template <> HD Fp f_zero<Fp>() { return fp_zero(); }
template <> HD Fp f_one<Fp>() { return fp_one(); }
template <> HD Fp2 f_zero<Fp2>() { return fp2_zero(); }
template <> HD Fp2 f_one<Fp2>() { return fp2_one(); }
template <> HD FpI f_zero<FpI>() { return FpI{fp_zero()}; }
template <> HD FpI f_one<FpI>() { return FpI{fp_one()}; }
template <class F> HD F curve_b();
template <> HD Fp curve_b<Fp>() { return Fp{{K_B1}}; }
template <> HD FpI curve_b<FpI>() { return FpI{Fp{{K_B1}}}; }
template <> HD Fp2 curve_b<Fp2>() { return Fp2{Fp{{K_B2_C0}}, Fp{{K_B2_C1}}}; }

}  // namespace eip
