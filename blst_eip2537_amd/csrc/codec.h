// EIP-2537 wire codec, shared by host and device.  Inputs are viewed as 32-bit words exactly
// as they sit in memory (little-endian loads of big-endian data), so one Fp element is 16
// words: 4 that must be zero, then 12 big-endian words.  Follows the reference's decode rules:
//   fp_from_bytes      reference src/eip2537.c:263-309  (-1 invalid / 0 zero / 1 non-zero)
//   decode_g1_point    reference src/eip2537.c:320-343
//   decode_g2_point    reference src/eip2537.c:381-404
//   decode_scalar      reference src/eip2537.c:417-420  (never fails, not reduced)
//   fp_to_bytes etc.   reference src/eip2537.c:312-317, 346-350, 371-378, 407-411
#pragma once
#include "curve.h"

namespace eip {

enum : int {
    E_SUCCESS = 0, E_NOT_ON_CURVE = 1, E_NOT_IN_SUBGROUP = 2, E_INVALID_ELEMENT = 3,
    E_ENCODING_ERROR = 4, E_INVALID_LENGTH = 5, E_EMPTY_INPUT = 6, E_MEMORY_ERROR = 7
};

HD uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// -1 invalid, 0 zero, 1 non-zero; `out` is in Montgomery form on success
HD int fp_decode(Fp &out, const uint32_t *w) {
    const Fp p = fp_p();
    uint32_t pad = w[0] | w[1] | w[2] | w[3];
    Fp raw;
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        raw.l[11 - k] = bswap32(w[4 + k]);
        nz |= w[4 + k];
    }
    uint32_t borrow = 0;     // raw < p  <=>  raw - p borrows
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)raw.l[i] - p.l[i] - borrow;
        borrow = (uint32_t)(s >> 32) & 1u;
    }
    if (pad != 0 || borrow == 0) return -1;
    out = fp_to_mont(raw);
    return nz != 0;
}
HD void fp_encode(uint32_t *w, const Fp &a) {
    Fp raw = fp_from_mont(a);
    w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) w[4 + k] = bswap32(raw.l[11 - k]);
}
HD int fp_decode(Fp2 &out, const uint32_t *w) {
    int s0 = fp_decode(out.c0, w);
    int s1 = fp_decode(out.c1, w + 16);
    if (s0 < 0 || s1 < 0) return -1;
    return s0 | s1;
}
HD void fp_encode(uint32_t *w, const Fp2 &a) {
    fp_encode(w, a.c0);
    fp_encode(w + 16, a.c1);
}

template <class F> struct Wire;
template <> struct Wire<Fp> {      // G1
    static constexpr int kCoordWords = 16, kPointWords = 32, kMsmRecWords = 40;
};
template <> struct Wire<Fp2> {     // G2
    static constexpr int kCoordWords = 32, kPointWords = 64, kMsmRecWords = 72;
};

// Both coordinates are decoded before the verdict, so INVALID_ELEMENT outranks NOT_ON_CURVE;
// (0,0) is infinity and skips the curve test.  No subgroup check (reference :340,:401).
template <class F> HD int decode_point(Aff<F> &out, const uint32_t *w) {
    int sx = fp_decode(out.x, w);
    int sy = fp_decode(out.y, w + Wire<F>::kCoordWords);
    if (sx < 0 || sy < 0) return E_INVALID_ELEMENT;
    if (sx == 0 && sy == 0) { out.x = f_zero<F>(); out.y = f_zero<F>(); return E_SUCCESS; }
    if (!on_curve(out)) return E_NOT_ON_CURVE;
    return E_SUCCESS;
}
#if defined(__HIP_DEVICE_COMPILE__)
// The same decode with every product INLINED, for the decode kernels: a call of the out-of-line Fp product costs the
// calling kernel 160 B of scratch per lane for the callee's saved registers (k_msm_decode<Fp2>, k_msm_decode_batch and
// k_pair_decode carried it through round 2).  Same verdicts in the same order as decode_point().
__device__ __forceinline__ int fp_decode_inl(Fp &out, const uint32_t *w) {
    const Fp p = fp_p();
    const uint32_t pad = w[0] | w[1] | w[2] | w[3];
    Fp raw;
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        raw.l[11 - k] = bswap32(w[4 + k]);
        nz |= w[4 + k];
    }
    uint32_t borrow = 0;     // raw < p  <=>  raw - p borrows
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t s = (uint64_t)raw.l[i] - p.l[i] - borrow;
        borrow = (uint32_t)(s >> 32) & 1u;
    }
    if (pad != 0 || borrow == 0) return -1;
    out = fp_mul_cols(raw, Fp{{K_RR}});
    return nz != 0;
}
__device__ __forceinline__ int fp_decode_inl(Fp2 &out, const uint32_t *w) {
    const int s0 = fp_decode_inl(out.c0, w), s1 = fp_decode_inl(out.c1, w + 16);
    if (s0 < 0 || s1 < 0) return -1;
    return s0 | s1;
}
__device__ __forceinline__ bool on_curve_inl(const Aff<Fp> &a) {
    return eq(fp_sqr_cols(a.y), add(fp_mul_cols(fp_sqr_cols(a.x), a.x), curve_b<Fp>()));
}
__device__ __forceinline__ bool on_curve_inl(const Aff<Fp2> &a) {
    return eq(fp2_sqr_body(a.y), add(fp2_mul_body(fp2_sqr_body(a.x), a.x), curve_b<Fp2>()));
}
template <class F> __device__ __forceinline__ int decode_point_inl(Aff<F> &out, const uint32_t *w) {
    const int sx = fp_decode_inl(out.x, w), sy = fp_decode_inl(out.y, w + Wire<F>::kCoordWords);
    if (sx < 0 || sy < 0) return E_INVALID_ELEMENT;
    if (sx == 0 && sy == 0) { out.x = f_zero<F>(); out.y = f_zero<F>(); return E_SUCCESS; }
    if (!on_curve_inl(out)) return E_NOT_ON_CURVE;
    return E_SUCCESS;
}
#else
template <class F> HD int decode_point_inl(Aff<F> &out, const uint32_t *w) { return decode_point<F>(out, w); }     // the host pass only parses the kernels
#endif
template <class F> HD void encode_point(uint32_t *w, const Aff<F> &a) {
    fp_encode(w, a.x);
    fp_encode(w + Wire<F>::kCoordWords, a.y);
}
// 32 big-endian bytes -> 8 little-endian words
HD void decode_scalar(uint32_t k[8], const uint32_t *w) {
#pragma unroll
    for (int i = 0; i < 8; i++) k[7 - i] = bswap32(w[i]);
}

}  // namespace eip
