// Host check of csrc/limb30.h (the limb-form arithmetic of k_msm_accum_l): the same HD code, compiled for
// the host, against the 64-bit host product and the generic madd() of curve.h.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc
//         tools/limb30_check.hip -o limb30_check && ./limb30_check
#include <stdio.h>
#include <hip/hip_runtime.h>
#include "pairing.h"
#include "limb30.h"
using namespace eip;
static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static Fp rnd_fp() {                       // canonical, Montgomery form of a random value
    Fp v;
    for (int k = 0; k < 12; k++) v.l[k] = (uint32_t)rnd();
    v.l[11] &= 0x0fffffffu;
    return mul(v, fp_one());               // a product result is canonical
}
static Fp canon_of(const FpL &a) { return fp_reduce_once(to_fpi(a).v); }    // a R' -> a R canonical (the host fp_canon is the identity)
int main() {
    long bad = 0;
    const Fp r390{{K_R390_MODP}};
    for (int it = 0; it < 200000; it++) {
        const Fp a = rnd_fp(), b = rnd_fp();
        const FpL A = fpl_from_mont(a), B = fpl_from_mont(b);
        if (!eq(canon_of(A), a)) bad++;
        if (!eq(canon_of(mulL(A, B)), mul(a, b))) bad++;
        if (!eq(canon_of(sqrL(A)), sqr(a))) bad++;
        if (!eq(canon_of(subL<2>(A, B)), sub(a, b))) bad++;
        if (!eq(canon_of(sub2L<4>(A, B)), sub(a, dbl(b)))) bad++;
        if (!eq(canon_of(addL(A, B)), add(a, b))) bad++;
        if (!eq(canon_of(dbl_addL(A, B)), add(dbl(a), b))) bad++;
        // grown operands: (a + 8p - b) (b + 6p - a), bounds 9 x 7 < 630
        const FpL g1 = subL<8>(A, B), g2 = subL<6>(B, A);
        if (!eq(canon_of(mulL(g1, g2)), mul(sub(a, b), sub(b, a)))) bad++;
        if (!eq(canon_of(sqrL(g1)), sqr(sub(a, b)))) bad++;
        if (!eq(canon_of(mul2L(g1, g2, A, negL<2>(B))), sub(mul(sub(a, b), sub(b, a)), mul(a, b)))) bad++;
        // six products, one reduction (round 4: the fused line product of the pairing fold)
        if (!eq(canon_of(mul6L(g1, g2, A, negL<2>(B), A, A, B, g1, g2, g2, A, B)),
                add(add(sub(mul(sub(a, b), sub(b, a)), mul(a, b)), add(mul(a, a), mul(b, sub(a, b)))), add(mul(sub(b, a), sub(b, a)), mul(a, b))))) bad++;
        if (!eq(canon_of(mul4L(g1, g2, A, negL<2>(B), A, A, B, g1)),
                add(sub(mul(sub(a, b), sub(b, a)), mul(a, b)), add(mul(a, a), mul(b, sub(a, b)))))) bad++;
        if (is_zero_modp(g1, 10) != eq(a, b)) bad++;
        if (!is_zero_modp(subL<8>(A, A), 10) || !is_zero_modp(subL<3>(A, A), 4)) bad++;
    }
    // the two-product sum with one reduction of the 12 x 32-bit world (field.h, fp_mul2_cols30): operands in [0, 2p)
    for (int it = 0; it < 300000; it++) {
        Fp v[4];
        for (auto &x : v) { for (int k = 0; k < 12; k++) x.l[k] = (uint32_t)rnd(); x.l[11] &= 0x0fffffffu; }      // < 2^380 < 2p
        if (it % 7 == 0) for (int k = 0; k < 12; k++) v[it % 4].l[k] = k == 11 ? 0x0fffffffu : 0xffffffffu;
        const Fp got = fp_reduce_once(fp_mul2_cols30(v[0], v[1], v[2], v[3]));
        const Fp want = add(mul(fp_reduce_once(v[0]), fp_reduce_once(v[1])), mul(fp_reduce_once(v[2]), fp_reduce_once(v[3])));
        if (!eq(got, want)) bad++;
    }
    printf("field operations: %ld mismatches\n", bad);
    // extreme limbs: x = all-ones limbs (< 2^384 < 10 p).  With v = x mod p (plain words), mul(v, v) of the R
    // world is x^2 / R; the limb square gives x^2 / R', to_fpi turns that into x^2 R / R'^2, and two products
    // with 2^390 mod p (each: times 2^390 / R) bring it to x^2 / R as well.
    {
        FpL x;
        for (int k = 0; k < 12; k++) x.l[k] = kM30;
        x.l[12] = 0x00ffffffu;
        Fp v = from_limbs(x);
        for (int r = 0; r < 12; r++) v = fp_reduce_once(v);
        const Fp got = mul(mul(fp_reduce_once(to_fpi(sqrL(x)).v), r390), r390);
        // the same operand through the general product and the two-product sum (their own carry-out schedules)
        const Fp sq = canon_of(sqrL(x));
        const Fp sq6 = add(add(add(sq, sq), add(sq, sq)), add(sq, sq));
        const bool ok = eq(got, mul(v, v)) && eq(canon_of(mulL(x, x)), sq) && eq(canon_of(mul2L(x, x, x, x)), add(sq, sq)) &&
                        eq(canon_of(mul6L(x, x, x, x, x, x, x, x, x, x, x, x)), sq6) && eq(canon_of(mul4L(x, x, x, x, x, x, x, x)), add(add(sq, sq), add(sq, sq)));
        printf("extreme limbs: %s\n", ok ? "ok" : "MISMATCH");
        if (!ok) bad++;
    }
    // accumulate loop against madd(): random points are not needed -- the formulas are field identities,
    // so any (x, y) pairs exercise them; equal and opposite entries exercise the rare paths
    long bad2 = 0;
    for (int it = 0; it < 20000; it++) {
        Aff<Fp> pts[6];
        for (auto &q : pts) q = Aff<Fp>{rnd_fp(), rnd_fp()};
        Xyzz<Fp> ref = xyzz_inf<Fp>();
        AccL acc; bool inf = true;
        const int n = 1 + (int)(rnd() % 12);
        for (int e = 0; e < n; e++) {
            const uint64_t r = rnd();
            Aff<Fp> q = pts[r % 6];
            if (e == 1 && (r >> 8) % 3 == 0) q = pts[0];            // doubling of the first entry ...
            if ((r >> 16) & 1) q.y = neg(q.y);
            ref = madd(ref, q);
            const Fp yr = mul(q.y, r390);
            madd_l(acc, inf, fpl_from_mont(q.x), to_limbs(yr));
        }
        if (inf != is_inf(ref)) { bad2++; continue; }
        if (inf) continue;
        if (!eq(canon_of(acc.x), ref.x) || !eq(canon_of(acc.y), ref.y) || !eq(canon_of(acc.zz), ref.zz) || !eq(canon_of(acc.zzz), ref.zzz)) bad2++;
    }
    printf("accumulate chains: %ld mismatches\n", bad2);
    // complete additions and doublings of limb-form points (fold / reduce kernels) against curve.h
    long bad3 = 0;
    auto lift = [&](const Xyzz<Fp> &v) {
        return Xyzz<FpL>{fpl_from_mont(v.x), fpl_from_mont(v.y), fpl_from_mont(v.zz), fpl_from_mont(v.zzz)};
    };
    auto same = [&](const Xyzz<FpL> &l, const Xyzz<Fp> &v) {
        if (is_inf(v)) return is_zero(l.zz);
        const Xyzz<Fp> c = canon(l);
        return eq(c.x, v.x) && eq(c.y, v.y) && eq(c.zz, v.zz) && eq(c.zzz, v.zzz);
    };
    for (int it = 0; it < 20000; it++) {
        Xyzz<Fp> a{rnd_fp(), rnd_fp(), rnd_fp(), rnd_fp()}, b{rnd_fp(), rnd_fp(), rnd_fp(), rnd_fp()};
        const uint64_t r = rnd();
        if (r % 7 == 0) b = a;                                   // doubling through add
        if (r % 7 == 1) b = neg(a);                              // cancellation
        if (r % 7 == 2) a = xyzz_inf<Fp>();
        if (r % 7 == 3) b = xyzz_inf<Fp>();
        Xyzz<FpL> la = lift(a), lb = lift(b);
        Xyzz<Fp> ra = a;
        for (int step = 0; step < 6; step++) {                   // a chain, so that grown bounds feed back
            if (!same(la, ra)) { bad3++; break; }
            if ((r >> (8 + step)) & 1) { la = add(la, lb); ra = add(ra, b); }
            else { la = dbl(la); ra = dbl(ra); }
        }
        if (!same(la, ra)) bad3++;
    }
    printf("point operations: %ld mismatches\n", bad3);
    bad2 += bad3;
    return bad || bad2 ? 1 : 0;
}
