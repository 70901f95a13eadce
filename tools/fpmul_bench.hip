// Micro-benchmark: formulations of the 381-bit Montgomery product on gfx950.
//   V0  12 x 32-bit limbs, CIOS (csrc/field.h as shipped in round 1)
//   V2/V3  what csrc/field.h ships: V1's core between a 12x32 -> 14x28 unpack and a repack + reduce
//   V1  14 x 28-bit limbs, 64-bit column accumulators: every multiply-add is ONE v_mad_u64_u32
//       accumulating in place, no carry handling inside the loops (28 x 2^56 < 2^64)
// Each thread runs a dependent chain x = x*y; the grid fills the chip at 8 waves/SIMD if VGPRs allow.
//   hipcc -O3 --offload-arch=gfx950 -I blst_eip2537_amd/csrc tools/fpmul_bench.hip -o tools/fpmul_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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
    run<0>("V0 12x32 CIOS (inlined)", d);
    run<1>("V1 14x28 column accumulators", d);
    run<2>("V2 V1 as drop-in (12x32 in/out)", d);
    run<3>("V3 squaring, drop-in", d);
    run<4>("V4 13x30 columns, drop-in", d);
    run<5>("V5 13x30 squaring, drop-in", d);
    run<6>("V6 lazy product (library default)", d);
    return 0;
}
