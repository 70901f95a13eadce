// Host Montgomery product: ADX inline-assembly form against the portable mulx-rows form (csrc/field.h): equality on 2M random
// operands + edge values, then ns per product.  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc tools/host_mul_check.hip -o /tmp/host_mul_check
#include <stdio.h>
#include <chrono>
#include "pairing.h"
using namespace eip;
using namespace std::chrono;
static double now() { return duration<double>(steady_clock::now().time_since_epoch()).count(); }
#if defined(__HIP_DEVICE_COMPILE__)
int main() { return 0; }
#else
int main() {
    // random-ish operands incl. edge values; compare the ADX product with the portable one
    uint64_t s = 0x243F6A8885A308D3ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    long bad = 0;
    Fp pm1 = fp_p(); pm1.l[0] -= 1;
    Fp edge[4] = {fp_zero(), fp_one(), pm1, Fp{{1,0,0,0,0,0,0,0,0,0,0,0}}};
    for (int i = 0; i < 2000000; i++) {
        Fp a, b;
        for (int k = 0; k < 6; k++) { uint64_t v = next(), w = next(); memcpy(&a.l[2*k], &v, 8); memcpy(&b.l[2*k], &w, 8); }
        a.l[11] &= 0x0fffffff; b.l[11] &= 0x0fffffff;    // < 2^380 < p
        if (i < 16) { a = edge[i & 3]; b = edge[(i >> 2) & 3]; }
        Fp x = fp_mul_adx(a, b), y = fp_mul_limbs64(a, b);
        if (!eq(x, y)) bad++;
    }
    printf("mismatches: %ld\n", bad);
    Fp a = fp_to_mont(Fp{{3,5,7,11,13,17,19,23,29,31,37,1}}), b = fp_to_mont(Fp{{2,4,8,16,32,64,128,256,512,1024,2048,2}});
    int N = 5000000;
    double t0 = now(); for (int i = 0; i < N; i++) a = fp_mul_adx(a, b); double t1 = now();
    printf("adx      %.1f ns\n", (t1 - t0) / N * 1e9);
    t0 = now(); for (int i = 0; i < N; i++) a = fp_mul_limbs64(a, b); t1 = now();
    printf("limbs64  %.1f ns  %u\n", (t1 - t0) / N * 1e9, a.l[0]);
    return bad != 0;
}
#endif
