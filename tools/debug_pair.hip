#include <stdio.h>
#include <string.h>
#include "codec.h"
#include "pairing.h"
using namespace eip;
__global__ void __launch_bounds__(64) chk_g2(const uint32_t *__restrict__ in, uint32_t k, uint32_t *out) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= k) return;
    Aff<Fp2> q;
    int st = decode_point<Fp2>(q, in + (size_t)i * 96 + 32);
    uint32_t r = (uint32_t)st;
    if (st == E_SUCCESS) {
        r |= in_g2(q) ? 0x100u : 0u;
        Xyzz<Fp2> t = mul_zabs(q);
        Aff<Fp2> a = to_affine(t);
        out[64 + i * 2] = a.x.c0.l[0];
        out[64 + i * 2 + 1] = q.x.c0.l[0];
    }
    out[i] = r;
}
int main() {
    // host: encode k copies of (inf, G2)
    const int k = 3;
    Aff<Fp2> g2{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
    uint32_t rec[96 * k];
    memset(rec, 0, sizeof rec);
    for (int i = 0; i < k; i++) encode_point<Fp2>(rec + 96 * i + 32, g2);
    uint32_t *d_in, *d_out, h[256];
    hipMalloc(&d_in, sizeof rec); hipMalloc(&d_out, 1024);
    hipMemcpy(d_in, rec, sizeof rec, hipMemcpyHostToDevice);
    hipMemset(d_out, 0, 1024);
    chk_g2<<<1, 64>>>(d_in, k, d_out);
    hipMemcpy(h, d_out, 1024, hipMemcpyDeviceToHost);
    Aff<Fp2> ha = to_affine(mul_zabs(g2));
    for (int i = 0; i < k; i++) printf("lane %d: r=%x  zQ.x0=%08x (host %08x)  q.x0=%08x (host %08x)\n", i, h[i], h[64 + 2 * i], ha.x.c0.l[0], h[65 + 2 * i], g2.x.c0.l[0]);
    printf("host in_g2=%d\n", (int)in_g2(g2));
    return 0;
}
