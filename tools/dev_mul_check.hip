#include <stdio.h>
#include <hip/hip_runtime.h>
#include "pairing.h"
using namespace eip;
#if defined(__HIP_DEVICE_COMPILE__)
#define OPQ(x) asm volatile("" : "+v"(x))
#else
#define OPQ(x) asm volatile("" : "+r"(x))
#endif
// Reproducer for the hipcc (ROCm 7.2, gfx950) code-generation problem recorded in field.h: the 13-limb
// product below is the library's fp_mul_cols30_t WITHOUT the EIP_OPAQUE() lines.
//   hipcc -O3 -DVAR=0 ... : last reduction factor visible as "x & 0xffffff" -> ~99.6 % wrong on the device
//   hipcc -O3 -DVAR=1 ... : the factor hidden behind an empty asm           -> 0 wrong
// Build: hipcc -O3 -DVAR=0 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx
//        -Iblst_eip2537_amd/csrc tools/dev_mul_check.hip -o dev_mul_check
#if VAR == 1
#define BAR_M OPQ(m);
#else
#define BAR_M
#endif
HD Fp dbg30(const Fp &a, const Fp &b) {
    const bool REDUCE = true;
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
        BAR_M
#pragma unroll
        for (int j = 0; j < 13; j++) col[i + j] += (uint64_t)m * p30[j];
        if (i < 12) col[i + 1] += col[i] >> 30;
        if (i == 7) {
#pragma unroll
            for (int c = 8; c <= 16; c++) { col[c + 1] += col[c] >> 30; col[c] &= (uint64_t)M30; }
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

__global__ void k(const Fp *a, const Fp *b, int n, unsigned *bad) {
    int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Fp x = dbg30(a[i], b[i]);
    Fp y = fp_mul_cols28_t<true>(a[i], b[i]);
    if (!eq(x, y)) atomicAdd(&bad[0], 1u);
}
int main() {
    const int n = 1 << 14;
    Fp *ha = new Fp[n], *hb = new Fp[n];
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; i++) for (int k2 = 0; k2 < 12; k2++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; ha[i].l[k2] = (uint32_t)s; hb[i].l[k2] = (uint32_t)(s >> 32);
        if (k2 == 11) { ha[i].l[11] &= 0x0fffffff; hb[i].l[11] &= 0x0fffffff; }
    }
    Fp *da, *db; unsigned *dbad;
    hipMalloc(&da, n * sizeof(Fp)); hipMalloc(&db, n * sizeof(Fp)); hipMalloc(&dbad, 8);
    hipMemcpy(da, ha, n * sizeof(Fp), hipMemcpyHostToDevice); hipMemcpy(db, hb, n * sizeof(Fp), hipMemcpyHostToDevice); hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, da, db, n, dbad);
    unsigned bad[2];
    hipMemcpy(bad, dbad, 8, hipMemcpyDeviceToHost);
    printf("VAR %d: device mul mismatches %u of %d\n", VAR, bad[0], n);
    return 0;
}
