// Micro-benchmark: formulations of the 381-bit Montgomery product on gfx950.
//   V0  12 x 32-bit limbs, CIOS (csrc/field.h as shipped in round 1)
//   V2/V3  what csrc/field.h ships: V1's core between a 12x32 -> 14x28 unpack and a repack + reduce
//   V1  14 x 28-bit limbs, 64-bit column accumulators: every multiply-add is ONE v_mad_u64_u32
//       accumulating in place, no carry handling inside the loops (28 x 2^56 < 2^64)
// Each thread runs a dependent chain x = x*y; the grid fills the chip at 8 waves/SIMD if VGPRs allow.
//   hipcc -O3 -frounding-math --offload-arch=gfx950 -I blst_eip2537_amd/csrc tools/fpmul_bench.hip -o tools/fpmul_bench   (-frounding-math: V7's host check)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <fenv.h>
#include "field.h"
#include "fpmul_consts.h"
using namespace eip;

struct Fq { uint32_t l[14]; };
__device__ __forceinline__ Fq mul28(const Fq &a, const Fq &b) {
    const uint32_t p[14] = P28;
    const uint32_t M = (1u << 28) - 1u;
    uint64_t col[28];
#pragma unroll
    for (int i = 0; i < 28; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
#pragma unroll
        for (int j = 0; j < 14; j++) col[i + j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t m = ((uint32_t)col[i] * N0_28) & M;
#pragma unroll
        for (int j = 0; j < 14; j++) col[i + j] += (uint64_t)m * p[j];
        col[i + 1] += col[i] >> 28;
    }
    Fq r;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint64_t v = col[14 + k] + carry;
        r.l[k] = (uint32_t)v & M;
        carry = v >> 28;
    }
    return r;
}

// V7 (round 4, the K0 question of SURVEY.md 7.2): the FP64-FMA form.  8 limbs of 52 bits held as exact doubles; a partial product
// a_j b_i < 2^104 is split by two fused multiply-adds -- h = fma(a, b, 2^104) rounds it to a multiple of 2^52 (the high half, sitting in
// the mantissa of a double with a FIXED exponent; ROUND-TOWARD-ZERO mode, set once per kernel in the MODE register, makes that the floor),
// l = fma(a, b, (2^104 + 2^52) - h) is the exact remainder moved into [2^52, 2^53) (Emmart, Zheng, Weems 2018) -- and both halves are accumulated as the INTEGER bit patterns of those doubles in 64-bit columns (IEEE patterns are linear in the
// mantissa, also across the top of the binade), the known exponent patterns having been subtracted from the columns beforehand.
// Montgomery factor 2^416.  Per partial product: 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer additions, against ONE v_mad_u64_u32 per
// partial product of the 13 x 30-bit form (169 of them instead of 64, but 1 instruction each).
struct F52 { double l[8]; };
__host__ __device__ inline long long dbits(double d) { long long v; __builtin_memcpy(&v, &d, 8); return v; }
__host__ __device__ inline double bitsd(long long v) { double d; __builtin_memcpy(&d, &v, 8); return d; }
__host__ __device__ inline F52 mulfma(const F52 &a, const F52 &b) {
    const double p[8] = P52, n0 = N0_52;
    const double C1 = 0x1p104, C3 = 0x1p104 + 0x1p52;
    const long long BH = 0x4670000000000000ll, BX = 0x4330000000000000ll, BL = BX, M52 = (1ll << 52) - 1;
    long long col[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const int cl = k <= 14 ? (k < 14 - k ? k : 14 - k) + 1 : 0, ch = k >= 1 && k - 1 <= 14 ? (k - 1 < 15 - k ? k - 1 : 15 - k) + 1 : 0;
        col[k] = -(2ll * cl * BL + 2ll * ch * BH);             // every term's exponent pattern, removed beforehand (wraps mod 2^64)
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double h = __builtin_fma(a.l[j], b.l[i], C1), l = __builtin_fma(a.l[j], b.l[i], C3 - h);
            col[i + j + 1] += dbits(h);
            col[i + j] += dbits(l);
        }
        // m = column i * (-1 / p) mod 2^52 (the pending m p_0 term's pattern is still missing from the column: + BL)
        const double dv = bitsd(((col[i] + BL) & M52) | BX) - 0x1p52;
        const double h2 = __builtin_fma(dv, n0, C1), l2 = __builtin_fma(dv, n0, C3 - h2);
        const double dm = bitsd(((dbits(l2) - BL) & M52) | BX) - 0x1p52;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double h = __builtin_fma(dm, p[j], C1), l = __builtin_fma(dm, p[j], C3 - h);
            col[i + j + 1] += dbits(h);
            col[i + j] += dbits(l);
        }
        col[i + 1] += col[i] >> 52;                             // column i is now a multiple of 2^52
    }
    F52 r;
    long long c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const long long v = col[8 + k] + c;
        r.l[k] = bitsd((k < 7 ? (v & M52) : v) | BX) - 0x1p52;
        c = v >> 52;
    }
    return r;
}

template <int V> __global__ void __launch_bounds__(256) chain(uint32_t *out, int iters, uint32_t salt) {
    uint32_t acc = 0;
    if (V == 0) {
        Fp x{A32}, y{B32};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = fp_mul_limbs32(x, y);
        for (int i = 0; i < 12; i++) acc ^= x.l[i];
    } else if (V == 2) {
        Fp x{A32}, y{B32};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = fp_mul_cols28(x, y);
        for (int i = 0; i < 12; i++) acc ^= x.l[i];
    } else if (V == 3) {
        Fp x{A32};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = fp_sqr_cols28(x);
        for (int i = 0; i < 12; i++) acc ^= x.l[i];
    } else if (V == 4) {
        Fp x{A32}, y{B32};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = fp_mul_cols30_t<true>(x, y);
        for (int i = 0; i < 12; i++) acc ^= x.l[i];
    } else if (V == 5) {
        Fp x{A32};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = fp_sqr_cols30_t<true>(x);
        for (int i = 0; i < 12; i++) acc ^= x.l[i];
    } else if (V == 6) {
        FpI x{Fp{A32}}, y{Fp{B32}};
        x.v.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = mul(x, y);
        for (int i = 0; i < 12; i++) acc ^= x.v.l[i];
    } else if (V == 7) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);          // MODE.FP_ROUND (f64 / f16) = toward zero
#endif
        F52 x{A52}, y{B52};
        x.l[0] += (double)((threadIdx.x + salt) & 0xff);
        for (int i = 0; i < iters; i++) x = mulfma(x, y);
        for (int i = 0; i < 8; i++) acc ^= (uint32_t)dbits(x.l[i]) ^ (uint32_t)(dbits(x.l[i]) >> 32);
    } else {
        Fq x{A28}, y{B28};
        x.l[0] ^= (threadIdx.x + salt) & 0xff;
        for (int i = 0; i < iters; i++) x = mul28(x, y);
        for (int i = 0; i < 14; i++) acc ^= x.l[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void check(uint32_t *out) {
    Fp x{A32}, y{B32};
    Fp r = fp_mul_limbs32(x, y);
    for (int i = 0; i < 12; i++) out[i] = r.l[i];
    Fq u{A28}, v{B28};
    Fq s = mul28(u, v);
    for (int i = 0; i < 14; i++) out[16 + i] = s.l[i];
}
template <int V> static void run(const char *name, uint32_t *d) {
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chain<V><<<blocks, 256>>>(d, iters, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) chain<V><<<blocks, 256>>>(d, iters, 2 + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double muls = 3.0 * blocks * 256.0 * iters;
    printf("%-34s %8.2f G Fp-products/s   (%.2f ms per launch)\n", name, muls / (ms * 1e-3) / 1e9, ms / 3);
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    check<<<1, 1>>>(d);
    uint32_t h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const uint32_t r32[12] = R32, r28[14] = R28, p28[14] = P28;
    int ok32 = 1, ok28 = 1;
    for (int i = 0; i < 12; i++) ok32 &= h[i] == r32[i];
    // V1 returns a value in [0, 2p): accept r or r + p
    unsigned long long c = 0; int okp = 1;
    for (int i = 0; i < 14; i++) { ok28 &= h[16 + i] == r28[i]; unsigned long long s = (unsigned long long)r28[i] + p28[i] + c; okp &= h[16 + i] == (uint32_t)(s & 0xfffffff); c = s >> 28; }
    printf("check: V0 %s, V1 %s\n", ok32 ? "ok" : "MISMATCH", (ok28 || okp) ? "ok" : "MISMATCH");
    {   // V7 on the host (plain doubles and fma): a b 2^-416 mod p, or that plus p
        const F52 u{A52}, v{B52}, want{R52};
        fesetround(FE_TOWARDZERO);
        const F52 got = mulfma(u, v);
        fesetround(FE_TONEAREST);
        const unsigned long long pi[8] = P52_INT;
        int ok = 1, okq = 1;
        unsigned long long cc = 0;
        for (int i = 0; i < 8; i++) {
            ok &= got.l[i] == want.l[i];
            const unsigned long long sres = (unsigned long long)want.l[i] + pi[i] + cc;
            okq &= (unsigned long long)got.l[i] == (i < 7 ? (sres & ((1ull << 52) - 1)) : sres);
            cc = sres >> 52;
        }
        printf("check: V7 (host) %s\n", (ok || okq) ? "ok" : "MISMATCH");
    }
    run<0>("V0 12x32 CIOS (inlined)", d);
    run<1>("V1 14x28 column accumulators", d);
    run<2>("V2 V1 as drop-in (12x32 in/out)", d);
    run<3>("V3 squaring, drop-in", d);
    run<4>("V4 13x30 columns, drop-in", d);
    run<5>("V5 13x30 squaring, drop-in", d);
    run<6>("V6 lazy product (library default)", d);
    run<7>("V7 8x52 FP64-FMA columns", d);
    return 0;
}
