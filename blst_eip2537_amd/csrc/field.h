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

// r = t - p if t >= p else t, for t < 2p (t fits in 384 bits)
HD Fp fp_reduce_once(const Fp &t) {
    const Fp p = fp_p();
    Fp d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)t.l[i] - p.l[i] - borrow;
        d.l[i] = (uint32_t)s;
        borrow = (uint32_t)(s >> 32) & 1u;
    }
    uint32_t keep_t = 0u - borrow;   // all ones when t < p
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = (t.l[i] & keep_t) | (d.l[i] & ~keep_t);
    return r;
}
HD Fp add(const Fp &a, const Fp &b) {
    Fp t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
        t.l[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    return fp_reduce_once(t);     // a + b < 2p < 2^384: no carry out
}
HD Fp sub(const Fp &a, const Fp &b) {
    const Fp p = fp_p();
    Fp d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)s;
        borrow = (uint32_t)(s >> 32) & 1u;
    }
    uint32_t mask = 0u - borrow, c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)d.l[i] + (p.l[i] & mask) + c;
        r.l[i] = (uint32_t)s;
        c = (uint32_t)(s >> 32);
    }
    return r;
}
HD Fp neg(const Fp &a) { return sub(fp_zero(), a); }
HD Fp dbl(const Fp &a) { return add(a, a); }

// Montgomery product, 12 x 32-bit CIOS.  Each inner step is one v_mad_u64_u32 plus carry adds.
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

#if !defined(__HIP_DEVICE_COMPILE__)
// Host: the same 48 bytes as 6 x 64-bit limbs.
inline Fp fp_mul_limbs64(const Fp &a, const Fp &b) {
    typedef unsigned __int128 u128;
    static const uint32_t pw[12] = {K_P};
    uint64_t A[6], B[6], Pm[6], t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(A, a.l, 48);
    memcpy(B, b.l, 48);
    memcpy(Pm, pw, 48);
    for (int i = 0; i < 6; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 6; j++) {
            u128 s = (u128)A[j] * B[i] + t[j] + c;
            t[j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        t[6] = c;
        uint64_t m = t[0] * K_N0_64;
        u128 s = (u128)m * Pm[0] + t[0];
        c = (uint64_t)(s >> 64);
        for (int j = 1; j < 6; j++) {
            s = (u128)m * Pm[j] + t[j] + c;
            t[j - 1] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        t[5] = t[6] + c;
    }
    Fp r;
    memcpy(r.l, t, 48);
    return fp_reduce_once(r);
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __noinline__ Fp fp_mul_outlined(Fp a, Fp b) { return fp_mul_limbs32(a, b); }
#endif

HD Fp mul(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fp_mul_outlined(a, b);
#else
    return fp_mul_limbs64(a, b);
#endif
}
HD Fp sqr(const Fp &a) { return mul(a, a); }

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
HD Fp2 mul(const Fp2 &a, const Fp2 &b) {
    Fp t0 = mul(a.c0, b.c0);
    Fp t1 = mul(a.c1, b.c1);
    Fp t2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fp2{sub(t0, t1), sub(sub(t2, t0), t1)};
}
HD Fp2 sqr(const Fp2 &a) {
    Fp m = mul(a.c0, a.c1);
    return Fp2{mul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(m)};
}
HD Fp2 mul_fp(const Fp2 &a, const Fp &s) { return Fp2{mul(a.c0, s), mul(a.c1, s)}; }
HD Fp2 mul_xi(const Fp2 &a) { return Fp2{sub(a.c0, a.c1), add(a.c0, a.c1)}; }   // * (1 + u)
HD Fp2 inv(const Fp2 &a) {
    Fp n = inv(add(sqr(a.c0), sqr(a.c1)));
    return Fp2{mul(a.c0, n), neg(mul(a.c1, n))};
}

// uniform spelling for the curve templates
template <class F> HD F f_zero();
template <class F> HD F f_one();
template <> HD Fp f_zero<Fp>() { return fp_zero(); }
template <> HD Fp f_one<Fp>() { return fp_one(); }
template <> HD Fp2 f_zero<Fp2>() { return fp2_zero(); }
template <> HD Fp2 f_one<Fp2>() { return fp2_one(); }
template <class F> HD F curve_b();
template <> HD Fp curve_b<Fp>() { return Fp{{K_B1}}; }
template <> HD Fp2 curve_b<Fp2>() { return Fp2{Fp{{K_B2_C0}}, Fp{{K_B2_C1}}}; }

}  // namespace eip
