// Host check of csrc/ifma.h (AVX-512 IFMA arithmetic of the pairing's host tail) against the scalar code of pairing.h.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc tools/ifma_check.hip -o /tmp/ifma_check
#include <stdio.h>
#include <chrono>
#include <hip/hip_runtime.h>
#include "pairing.h"
#include "ifma.h"
#include "h2c.h"
#include "ifma_horner.h"
using namespace eip;
static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static Fp rnd_fp() { Fp v; for (int k = 0; k < 12; k++) v.l[k] = (uint32_t)rnd(); v.l[11] &= 0x0fffffffu; return mul(v, fp_one()); }
static Fp2 r2() { return Fp2{rnd_fp(), rnd_fp()}; }
static Fp12 r12() { return Fp12{Fp6{r2(), r2(), r2()}, Fp6{r2(), r2(), r2()}}; }
#if defined(EIP_HAVE_IFMA)
__attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl,avx512bw"))) static int run() {
    using namespace ifma;
    long bad = 0;
    for (int it = 0; it < 20000; it++) {
        Fp a[8], b[8], out[8];
        const Fp *pa[8], *pb[8];
        Fp *po[8];
        for (int i = 0; i < 8; i++) {
            a[i] = rnd_fp(); b[i] = rnd_fp();
            if (it % 5 == 0 && i == 3) { a[i] = fp_zero(); }
            if (it % 7 == 0 && i == 5) { a[i] = sub(fp_zero(), fp_one()); b[i] = a[i]; }       // p - 1 (in Montgomery form: -1)
            pa[i] = &a[i]; pb[i] = &b[i]; po[i] = &out[i];
        }
        const V8 va = vload(pa), vb = vload(pb);
        vstore(po, va);
        for (int i = 0; i < 8; i++) if (!eq(out[i], a[i])) bad++;
        vstore(po, vmul(va, vb));
        for (int i = 0; i < 8; i++) if (!eq(out[i], mul(a[i], b[i]))) bad++;
        // lazy chains: (a + b)(a - b + 64 p), reduced
        vstore(po, vmul(vnorm(vadd(va, vb)), vnorm(vsub64(va, vb))));
        for (int i = 0; i < 8; i++) if (!eq(out[i], mul(add(a[i], b[i]), sub(a[i], b[i])))) bad++;
        V8 big = va;
        for (int k = 0; k < 9; k++) big = vadd(big, big);                 // 512 a
        vstore(po, vreduce(vnorm(big)));
        for (int i = 0; i < 8; i++) {
            Fp w = a[i];
            for (int k = 0; k < 9; k++) w = add(w, w);
            if (!eq(out[i], w)) bad++;
        }
    }
    printf("vector field operations: %ld mismatches\n", bad);
    long bad2 = 0;
    for (int it = 0; it < 300; it++) {
        Fp12 f = r12();
        Fp12 want = f;
        Cyc c = cyc_load(f);
        for (int k = 0; k < 40; k++) {
            want = cyclotomic_sqr(want);
            c = cyc_sqr(c);
            if (k % 13 == 0 && !eq(cyc_store(c), want)) { bad2++; break; }
        }
        if (!eq(cyc_store(c), want)) bad2++;
    }
    printf("cyclotomic squaring chains: %ld mismatches\n", bad2);
    long bad3 = 0;
    for (int it = 0; it < 400; it++) {
        const Fp12 f = r12(), g = r12();
        const F12v vf = f12_load(f), vg = f12_load(g);
        if (!eq(f12_store(vf), f)) bad3++;
        if (!eq(f12_store(f12_mul(vf, vg)), mul(f, g))) bad3++;
        if (!eq(f12_store(f12_mul(vf, vf)), sqr(f))) bad3++;
        if (!eq(f12_store(f12_conj(vf)), conj(f))) bad3++;
        if (!eq(f12_store(f12_mul(f12_conj(f12_mul(vf, vg)), f12_conj(vg))), mul(conj(mul(f, g)), conj(g)))) bad3++;    // chained, lazily reduced operands
        if (!eq(f12_store(from_cyc(cyc_sqr(to_cyc(f12_load(f))))), cyclotomic_sqr(f))) bad3++;
    }
    printf("Fp12 products: %ld mismatches\n", bad3);
    long bad4 = 0;
    for (int it = 0; it < 12; it++) {
        // an element of the cyclotomic subgroup: the easy part of the final exponentiation of a random element
        const Fp12 f = r12();
        const Fp12 f1 = mul(conj(f), inv(f)), g = mul(frob2(f1), f1);
        if (!eq(conj(f12_store(f12_exp_zabs(f12_load(g)))), exp_by_z(g))) bad4++;
        if (!eq(final_exp_ifma(f), final_exp(f))) bad4++;
        Fp12 L[17];
        int sq[17];
        Fp12 want = fp12_one();
        for (int k = 0; k < 17; k++) {
            L[k] = r12();
            sq[k] = k ? 1 + (int)(rnd() % 4) : 0;
            for (int e = 0; e < sq[k]; e++) want = sqr(want);
            want = k ? mul(want, L[k]) : L[0];
        }
        if (!eq(horner_groups_ifma(L, sq, 17), want)) bad4++;
    }
    printf("exponentiation by z, final exponentiation, Horner: %ld mismatches\n", bad4);
    // ---- doubling chains of the multiexp host tail (ifma_horner.h) against curve.h ----
    long bad5 = 0;
    {
        const Aff<Fp> g1{Fp{{K_G1_X}}, Fp{{K_G1_Y}}};
        const Aff<Fp2> g2{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
        auto same1 = [](const Xyzz<Fp> &a, const Xyzz<Fp> &b) {
            if (is_zero(a.zz) || is_zero(b.zz)) return is_zero(a.zz) && is_zero(b.zz);
            const Aff<Fp> x = to_affine(a), y = to_affine(b);
            return eq(x.x, y.x) && eq(x.y, y.y);
        };
        auto same2 = [](const Xyzz<Fp2> &a, const Xyzz<Fp2> &b) {
            if (is_zero(a.zz) || is_zero(b.zz)) return is_zero(a.zz) && is_zero(b.zz);
            const Aff<Fp2> x = to_affine(a), y = to_affine(b);
            return eq(x.x, y.x) && eq(x.y, y.y);
        };
        for (int it = 0; it < 200; it++) {
            uint32_t k[8];
            for (auto &w : k) w = (uint32_t)rnd();
            // a random XYZZ representative of a random multiple (every fifth: a curve point outside the subgroup)
            Xyzz<Fp> p1 = scalar_mul(g1, k, 256);
            Xyzz<Fp2> p2 = scalar_mul(g2, k, 256);
            if (it % 5 == 4) {
                const Aff<Fp> c1 = map_to_curve<Fp>(rnd_fp());
                const Aff<Fp2> c2 = map_to_curve<Fp2>(r2());
                p1 = Xyzz<Fp>{c1.x, c1.y, fp_one(), fp_one()};
                p2 = Xyzz<Fp2>{c2.x, c2.y, fp2_one(), fp2_one()};
            }
            const int n = 1 + (int)(rnd() % 17);
            Xyzz<Fp> w1 = p1, v1 = p1;
            Xyzz<Fp2> w2 = p2, v2 = p2;
            for (int i = 0; i < n; i++) { w1 = dbl(w1); w2 = dbl(w2); }
            double_n(v1, n);
            double_n(v2, n);
            if (!same1(v1, w1)) bad5++;
            if (!same2(v2, w2)) bad5++;
            // a whole Horner pass: 16 windows of 16 doublings with an addition between them
            if (it < 20) {
                Xyzz<Fp> a1 = xyzz_inf<Fp>(), b1 = a1;
                Xyzz<Fp2> a2 = xyzz_inf<Fp2>(), b2 = a2;
                for (int w = 0; w < 16; w++) {
                    for (int i = 0; i < 16; i++) { a1 = dbl(a1); a2 = dbl(a2); }
                    horner_double_n(b1, 16);
                    horner_double_n(b2, 16);
                    if (w % 5 != 3) { a1 = add(a1, p1); b1 = add(b1, p1); a2 = add(a2, p2); b2 = add(b2, p2); }
                }
                if (!same1(a1, b1) || !same2(a2, b2)) bad5++;
            }
        }
        // the vector Horner accumulator (G2Horner: doubling chains AND additions in one vector) against the scalar pass, with window
        // sums that are infinity, equal to the accumulator (doubling case) and opposite to it (cancellation)
        for (int it = 0; it < 60; it++) {
            uint32_t k[8];
            for (auto &w : k) w = (uint32_t)rnd();
            const Xyzz<Fp2> p2 = scalar_mul(g2, k, 256);
            const Aff<Fp2> c2 = map_to_curve<Fp2>(r2());
            const Xyzz<Fp2> q2{c2.x, c2.y, fp2_one(), fp2_one()};
            Xyzz<Fp2> a = xyzz_inf<Fp2>();
            HornerAcc<Fp2> hv;
            for (int w = 0; w < 24; w++) {
                const int n = 1 + (int)(rnd() % 13);
                for (int i = 0; i < n; i++) a = dbl(a);
                hv.dbl_n(n);
                Xyzz<Fp2> term = (w % 3 == 0) ? p2 : q2;
                const int kind = (it + w) % 7;
                if (kind == 1) term = xyzz_inf<Fp2>();
                if (kind == 2 && !is_zero(a.zz)) term = a;                                   // P + P
                if (kind == 3 && !is_zero(a.zz)) term = Xyzz<Fp2>{a.x, neg(a.y), a.zz, a.zzz};  // P - P
                if (kind == 4 && !is_zero(a.zz)) {                                            // the same point in another representative
                    const Fp2 l = r2(), ll = sqr(l), lll = mul(ll, l);
                    term = Xyzz<Fp2>{mul(a.x, ll), mul(a.y, lll), mul(a.zz, ll), mul(a.zzz, lll)};
                }
                a = add(a, term);
                hv.add(term);
                if (!same2(hv.result(), a)) { bad5++; break; }
            }
        }
        Xyzz<Fp2> i2 = xyzz_inf<Fp2>();
        double_n(i2, 5);                                       // infinity stays infinity through the vectors too
        if (!is_zero(i2.zz)) bad5++;
        Xyzz<Fp> i1 = xyzz_inf<Fp>();
        double_n(i1, 5);
        if (!is_zero(i1.zz)) bad5++;
    }
    printf("multiexp doubling chains: %ld mismatches\n", bad5);
    {
        const Aff<Fp> g1{Fp{{K_G1_X}}, Fp{{K_G1_Y}}};
        const Aff<Fp2> g2{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
        Xyzz<Fp> a1{g1.x, g1.y, fp_one(), fp_one()}, b1 = a1;
        Xyzz<Fp2> a2{g2.x, g2.y, fp2_one(), fp2_one()}, b2 = a2;
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2560; i++) a1 = dbl(a1);
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 160; i++) double_n(b1, 16);
        auto t2 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2560; i++) a2 = dbl(a2);
        auto t3 = std::chrono::steady_clock::now();
        for (int i = 0; i < 160; i++) double_n(b2, 16);
        auto t4 = std::chrono::steady_clock::now();
        printf("doubling (chains of 16, conversions included): G1 scalar %.3f us, ifma %.3f us; G2 scalar %.3f us, ifma %.3f us (%d)\n", us(t0, t1) / 2560, us(t1, t2) / 2560,
               us(t2, t3) / 2560, us(t3, t4) / 2560, (int)((a1.x.l[0] ^ b1.x.l[0] ^ a2.x.c0.l[0] ^ b2.x.c0.l[0]) & 1));
    }
    // timing
    Fp12 f = r12();
    auto t0 = std::chrono::steady_clock::now();
    Fp12 x = f;
    for (int i = 0; i < 50; i++) x = final_exp(x);
    auto t1 = std::chrono::steady_clock::now();
    Fp12 y = f;
    for (int i = 0; i < 50; i++) y = final_exp_ifma(y);
    auto t2 = std::chrono::steady_clock::now();
    F12v v = f12_load(f);
    for (int i = 0; i < 2000; i++) v = f12_mul(v, v);
    auto t3 = std::chrono::steady_clock::now();
    Cyc c = to_cyc(f12_load(f));
    for (int i = 0; i < 2000; i++) c = cyc_sqr(c);
    auto t4 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    printf("final_exp: scalar %.1f us, ifma %.1f us (%s); vector Fp12 product %.3f us, cyclotomic squaring %.3f us (%d)\n", us(t0, t1) / 50, us(t1, t2) / 50,
           eq(x, y) ? "equal" : "DIFFERENT", us(t2, t3) / 2000, us(t3, t4) / 2000, (int)(_mm512_reduce_add_epi64(v.c0.l[0]) + _mm512_reduce_add_epi64(c.c0.l[0])) & 1);
    return (bad || bad2 || bad3 || bad4 || bad5 || !eq(x, y)) ? 1 : 0;
}
#else
static int run() { return 0; }
namespace eip { namespace ifma { static bool cpu_has_ifma() { return false; } } }
#endif
int main() {
    if (!ifma::cpu_has_ifma()) { printf("no AVX-512 IFMA on this CPU: skipped\n"); return 0; }
    return run();
}
