// Host-side field arithmetic timings (Fp .. Fp12 products, final exponentiation): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc tools/host_field_bench.hip
#include <chrono>
#include <stdio.h>
#include "pairing.h"
using namespace eip;
using namespace std::chrono;
static double now() { return duration<double>(steady_clock::now().time_since_epoch()).count(); }
int main() {
    Fp a = fp_to_mont(Fp{{3,5,7,11,13,17,19,23,29,31,37,1}});
    Fp b = fp_to_mont(Fp{{2,4,8,16,32,64,128,256,512,1024,2048,2}});
    int N = 2000000;
    double t0 = now();
    for (int i = 0; i < N; i++) { a = mul(a, b); }
    double t1 = now();
    printf("fp mul  %.1f ns\n", (t1 - t0) / N * 1e9);
    t0 = now();
    for (int i = 0; i < N; i++) { a = add(a, b); b = sub(b, a); }
    t1 = now();
    printf("fp add+sub %.1f ns (pair)\n", (t1 - t0) / N * 1e9);
    Fp2 x{a, b}, y{b, a};
    t0 = now();
    for (int i = 0; i < N; i++) { x = mul(x, y); }
    t1 = now();
    printf("fp2 mul %.1f ns\n", (t1 - t0) / N * 1e9);
    Fp12 f;
    Fp2 *fp = reinterpret_cast<Fp2 *>(&f);
    for (int i = 0; i < 6; i++) { fp[i] = x; x = mul(x, y); }
    Fp12 g = f;
    int M = 20000;
    t0 = now();
    for (int i = 0; i < M; i++) g = mul(g, f);
    t1 = now();
    printf("fp12 mul %.2f us\n", (t1 - t0) / M * 1e6);
    t0 = now();
    for (int i = 0; i < M; i++) g = sqr(g);
    t1 = now();
    printf("fp12 sqr %.2f us\n", (t1 - t0) / M * 1e6);
    t0 = now();
    Fp12 e;
    for (int i = 0; i < 20; i++) { e = final_exp(g); g = mul(g, e); }
    t1 = now();
    printf("final_exp %.1f us\n", (t1 - t0) / 20 * 1e6);
    printf("%u\n", reinterpret_cast<uint32_t*>(&e)[0]);
    return 0;
}
