// Host check of csrc/limbk.h + csrc/pairing_limb.h: the lane-group code of the pairing kernels, instantiated
// with all shares of a group in arrays (HostLanes4 / HostLanes8 / HostQuad), against pairing.h / curve.h.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Xarch_host -mbmi2 -Xarch_host -madx -Iblst_eip2537_amd/csrc
//         tools/pairing_limb_check.hip -o /tmp/pairing_limb_check && /tmp/pairing_limb_check
#include <stdio.h>
#include <vector>
#include <hip/hip_runtime.h>
#include "pairing.h"
#include "h2c.h"
#include "pairing_limb.h"
#include "g2_limb.h"
using namespace eip;
static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }
static Fp rnd_fp() {
    Fp v;
    for (int k = 0; k < 12; k++) v.l[k] = (uint32_t)rnd();
    v.l[11] &= 0x0fffffffu;
    return mul(v, fp_one());
}
static Fp2 rnd_fp2() { return Fp2{rnd_fp(), rnd_fp()}; }
static Fp canon_of(const FpL &a) { return fp_reduce_once(to_fpi(a).v); }      // a R' (< 600 p) -> a R canonical
static FpL lift(const Fp &a) { return fpl_from_mont(a); }
// a random limb-form value congruent to `a`, up to K p
template <int K> static FpL lift_big(const Fp &a) {
    FpL v = lift(a);
    const int extra = K > 1 ? (int)(rnd() % K) : 0;
    uint32_t kp[13];
    kp30<1>(kp);
    FpL p1;
    for (int k = 0; k < 13; k++) p1.l[k] = kp[k];
    for (int e = 0; e < extra; e++) v = addL(v, p1);
    return v;
}

template <int N> static LV<1, N> splat(const Fp &a) { return lv_const<1, N>(lift(a)); }
// component shares of an Fp2 value on the 8 lanes of a walk group / the 4 lanes of a quad
template <int N> static LV<1, N> split_q(const Fp2 &a) {
    LV<1, N> r;
    for (int i = 0; i < N; i++) r.l[i] = lift((i & 1) ? a.c1 : a.c0);
    return r;
}
template <int K, int N> static Fp2 join_q(const LV<K, N> &a, int pair = 0) { return Fp2{canon_of(a.l[2 * pair]), canon_of(a.l[2 * pair + 1])}; }
template <int K, int N> static bool lanes_agree(const LV<K, N> &a) {          // replicated values: congruent on every lane pair
    for (int i = 2; i < N; i++) if (!eq(canon_of(a.l[i]), canon_of(a.l[i & 1]))) return false;
    return true;
}

static Aff<Fp> g1_gen() { return Aff<Fp>{Fp{{K_G1_X}}, Fp{{K_G1_Y}}}; }
static Aff<Fp2> g2_gen() { return Aff<Fp2>{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}}; }
template <class F> static Aff<F> rnd_multiple(const Aff<F> &g) {
    uint32_t k[8];
    for (auto &w : k) w = (uint32_t)rnd();
    return to_affine(scalar_mul(g, k, 256));
}

static Fp12 from_quad(const Fp12Q<4> &f) {
    // c = 0 lanes (0, 1): own[t] = f_{2t};  c = 1 lanes (2, 3): own[t] = f_{2((t+1)%3)+1}
    Fp2 w[6];
    for (int t = 0; t < 3; t++) {
        w[2 * t] = Fp2{canon_of(f.own[t].l[0]), canon_of(f.own[t].l[1])};
        w[2 * ((t + 1) % 3) + 1] = Fp2{canon_of(f.own[t].l[2]), canon_of(f.own[t].l[3])};
    }
    // tower: c0 = (w0, w2, w4), c1 = (w1, w3, w5)
    return Fp12{Fp6{w[0], w[2], w[4]}, Fp6{w[1], w[3], w[5]}};
}
static Fp12Q<4> to_quad(const Fp12 &a) {
    const Fp2 w[6] = {a.c0.a0, a.c1.a0, a.c0.a1, a.c1.a1, a.c0.a2, a.c1.a2};
    Fp12Q<4> f;
    for (int t = 0; t < 3; t++) {
        const Fp2 e = w[2 * t], o = w[2 * ((t + 1) % 3) + 1];
        f.own[t].l[0] = lift_big<TreeK::F>(e.c0); f.own[t].l[1] = lift_big<TreeK::F>(e.c1);
        f.own[t].l[2] = lift_big<TreeK::F>(o.c0); f.own[t].l[3] = lift_big<TreeK::F>(o.c1);
    }
    return f;
}
static Fp12 rnd_fp12() { return Fp12{Fp6{rnd_fp2(), rnd_fp2(), rnd_fp2()}, Fp6{rnd_fp2(), rnd_fp2(), rnd_fp2()}}; }
static void elem_from_fp12(uint32_t *e, const Fp12 &a, int kmax) {
    const Fp2 w[6] = {a.c0.a0, a.c1.a0, a.c0.a1, a.c1.a1, a.c0.a2, a.c1.a2};
    for (int k = 0; k < 6; k++) {
        FpL v0 = lift(w[k].c0), v1 = lift(w[k].c1);
        if (kmax == 3) { v0 = lift_big<3>(w[k].c0); v1 = lift_big<3>(w[k].c1); }
        elem_store(e, k, 0, v0);
        elem_store(e, k, 1, v1);
    }
}

int main() {
    long bad = 0;
    // ---- small helpers ----
    for (int it = 0; it < 100000; it++) {
        const Fp a = rnd_fp();
        const FpL big = lift_big<599>(a);
        const FpL w = weak_reduceL(big);
        if (!eq(canon_of(w), a)) bad++;
        // at most 3 p: w - 3 p must be negative, i.e. top limb comparison through subL on a 3 p bound is awkward; check 4 p - w >= 0 limbwise
        const FpL d = negL<3>(w);
        if ((int32_t)d.l[12] < 0) bad++;
        const FpL A = lift_big<20>(a);
        if (!eq(canon_of(shlL<3>(A)), dbl(dbl(dbl(a))))) bad++;
        if (!eq(canon_of(shlL<2>(A)), dbl(dbl(a)))) bad++;
        if (!eq(canon_of(shlL<1>(A)), dbl(a))) bad++;
    }
    {   // extreme: the largest legal operand of a weak reduction
        FpL x;
        for (int k = 0; k < 12; k++) x.l[k] = kM30;
        x.l[12] = kM30;
        const FpL w = weak_reduceL(x);
        const FpL d = negL<3>(w);
        if ((int32_t)d.l[12] < 0) bad++;
        if (!eq(canon_of(w), canon_of(weak_reduceL(w)))) bad++;
    }
    printf("helpers: %ld mismatches\n", bad);

    // ---- G1: projective doubling / addition on 4 lanes against curve.h ----
    long bad1 = 0;
    const HostLanes4 x4;
    auto g1_affine_eq = [&](const G1Pt<4> &p, const Xyzz<Fp> &want) {
        for (int i = 0; i < 4; i++) {
            const Fp X = canon_of(p.x.l[i]), Y = canon_of(p.y.l[i]), Z = canon_of(p.z.l[i]);
            if (is_inf(want)) { if (!is_zero(Z)) return false; continue; }
            if (is_zero(Z)) return false;
            const Aff<Fp> w = to_affine(want);
            if (!eq(mul(w.x, Z), X) || !eq(mul(w.y, Z), Y)) return false;
        }
        return true;
    };
    auto g1_lift = [&](const Aff<Fp> &a, bool big) {
        // a random projective representative with grown limbs
        const Fp z = rnd_fp();
        G1Pt<4> p;
        for (int i = 0; i < 4; i++) {
            p.x.l[i] = big ? lift_big<G1K::X>(mul(a.x, z)) : lift(mul(a.x, z));
            p.y.l[i] = big ? lift_big<G1K::Y>(mul(a.y, z)) : lift(mul(a.y, z));
            p.z.l[i] = big ? lift_big<G1K::Z>(z) : lift(z);
        }
        return p;
    };
    for (int it = 0; it < 3000; it++) {
        const Aff<Fp> a = (it & 1) ? rnd_multiple(g1_gen()) : map_to_curve<Fp>(rnd_fp());
        Aff<Fp> b = (it & 2) ? rnd_multiple(g1_gen()) : map_to_curve<Fp>(rnd_fp());
        if (it % 7 == 0) b = a;
        if (it % 7 == 1) b = neg(a);
        const G1Pt<4> pa = g1_lift(a, true), pb = g1_lift(b, true);
        const auto d = proj_dbl<PolFp4>(x4, pa.x, pa.y, pa.z);
        const G1Pt<4> dd{widen<G1K::X>(d.x), widen<G1K::Y>(d.y), widen<G1K::Z>(d.z)};
        if (!g1_affine_eq(dd, dbl(from_affine(a)))) bad1++;
        const G1Pt<4> s = proj_add_g1(x4, pa, pb);
        if (!g1_affine_eq(s, add(from_affine(a), from_affine(b)))) bad1++;
        // infinity operands: (0 : Y : 0)
        G1Pt<4> inf = pa;
        for (int i = 0; i < 4; i++) { inf.x.l[i] = fpl_zero(); inf.z.l[i] = fpl_zero(); }
        if (!g1_affine_eq(proj_add_g1(x4, inf, pb), from_affine(b))) bad1++;
        if (!g1_affine_eq(proj_add_g1(x4, pa, inf), from_affine(a))) bad1++;
        const auto di = proj_dbl<PolFp4>(x4, inf.x, inf.y, inf.z);
        if (!is_zero(canon_of(di.z.l[0])) || is_zero(canon_of(di.y.l[0]))) bad1++;
    }
    printf("G1 projective operations: %ld mismatches\n", bad1);
    // membership
    long bad2 = 0;
    for (int it = 0; it < 60; it++) {
        const Aff<Fp> a = (it % 3 == 0) ? rnd_multiple(g1_gen()) : map_to_curve<Fp>(rnd_fp());
        const bool want = in_g1(a);
        const LanePred<4> nm = g1_not_member_l(x4, splat<4>(a.x), splat<4>(a.y));
        for (int i = 0; i < 4; i++) if (nm.b[i] == want) bad2++;
    }
    {   // small-order points: [h1 * r / l] of a random curve point has order l (or 1) for l | h1
        // (cheap way to get them: multiply a random point by r, which leaves its cofactor part)
        const uint32_t r_words[8] = {K_R_ORDER};
        for (int it = 0; it < 6; it++) {
            const Aff<Fp> c = map_to_curve<Fp>(rnd_fp());
            const Aff<Fp> h = to_affine(scalar_mul(c, r_words, 255));          // in the cofactor subgroup
            if (is_inf(h)) continue;
            const LanePred<4> nm = g1_not_member_l(x4, splat<4>(h.x), splat<4>(h.y));
            if (nm.b[0] != !in_g1(h) || !nm.b[0]) bad2++;
        }
    }
    printf("G1 membership: %ld mismatches\n", bad2);

    // ---- Miller walk on 8 lanes: lines and running point against pairing.h ----
    long bad3 = 0;
    const HostLanes8 x8;
    const Fp m3 = neg(add(dbl(fp_one()), fp_one())), two = dbl(fp_one());
    for (int it = 0; it < 12; it++) {
        const Aff<Fp> P = rnd_multiple(g1_gen());
        const Aff<Fp2> Q = (it % 4 == 3) ? map_to_curve<Fp2>(rnd_fp2()) : rnd_multiple(g2_gen());
        const Fp xs = mul(P.x, m3), ys = mul(P.y, two);
        WalkPt<8> T;
        T.x = widen<WalkK::X>(split_q<8>(Q.x)); T.y = widen<WalkK::Y>(split_q<8>(Q.y)); T.z = widen<WalkK::Z>(split_q<8>(fp2_one()));
        const LV<1, 8> qx = split_q<8>(Q.x), qy = split_q<8>(Q.y);
        MillerT Tr{Q.x, Q.y, fp2_one()};
        Fp12 F = fp12_one(), Fr = fp12_one();
        const uint64_t z = K_Z_ABS;
        auto same_point = [&]() {                        // T (homogeneous) vs Tr (Jacobian): X / Z == Xr / Zr^2, Y / Z == Yr / Zr^3
            if (!lanes_agree(T.x) || !lanes_agree(T.y) || !lanes_agree(T.z)) return false;
            const Fp2 X = join_q(T.x), Y = join_q(T.y), Z = join_q(T.z);
            const Fp2 zz = sqr(Tr.z), zzz = mul(zz, Tr.z);
            return eq(mul(X, zz), mul(Tr.x, Z)) && eq(mul(Y, zzz), mul(Tr.y, Z));
        };
        for (int bit = 62; bit >= 0; bit--) {
            const LineRecD<8> l = miller_dbl_l(x8, T);
            // pair 0 stores a0, pair 1 a1, pair 2 a4 -- all pairs hold the same values here
            const Fp2 a0 = join_q(l.a0, 0), X = join_q(l.a1, 1), yz = join_q(l.a4, 2);
            F = mul_by_014(sqr(F), a0, mul_fp(sqr(X), xs), mul_fp(yz, ys));
            const Line lr = miller_dbl_step(Tr);
            Fr = mul_by_014(sqr(Fr), lr.a0, mul_fp(lr.a1, P.x), mul_fp(lr.a4, P.y));
            if (!same_point()) { bad3++; break; }
            if ((z >> bit) & 1ull) {
                const LineRecA<8> la = miller_add_l(x8, T, qx, qy);
                F = mul_by_014(F, join_q(la.a0, 0), mul_fp(join_q(la.a1, 1), xs), mul_fp(join_q(la.a4, 2), ys));
                const Line lq = miller_add_step(Tr, Q);
                Fr = mul_by_014(Fr, lq.a0, mul_fp(lq.a1, P.x), mul_fp(lq.a4, P.y));
                if (!same_point()) { bad3++; break; }
            }
        }
        if (!eq(final_exp(conj(F)), final_exp(conj(Fr)))) bad3++;
        const LanePred<8> nm = g2_not_member_l(x8, T, qx, qy);
        for (int i = 0; i < 8; i++) if (nm.b[i] == in_g2(Q)) bad3++;
    }
    {   // a Q of small order (cofactor part of a random twist point): the walk may hit T = +-Q; the verdict must still be "not a member"
        const uint32_t r_words[8] = {K_R_ORDER};
        for (int it = 0; it < 4; it++) {
            const Aff<Fp2> c = map_to_curve<Fp2>(rnd_fp2());
            const Aff<Fp2> Q = to_affine(scalar_mul(c, r_words, 255));
            if (is_inf(Q)) continue;
            WalkPt<8> T;
            T.x = widen<WalkK::X>(split_q<8>(Q.x)); T.y = widen<WalkK::Y>(split_q<8>(Q.y)); T.z = widen<WalkK::Z>(split_q<8>(fp2_one()));
            const LV<1, 8> qx = split_q<8>(Q.x), qy = split_q<8>(Q.y);
            const uint64_t z = K_Z_ABS;
            for (int bit = 62; bit >= 0; bit--) {
                (void)miller_dbl_l(x8, T);
                if ((z >> bit) & 1ull) (void)miller_add_l(x8, T, qx, qy);
            }
            const LanePred<8> nm = g2_not_member_l(x8, T, qx, qy);
            for (int i = 0; i < 8; i++) if (nm.b[i] == in_g2(Q)) bad3++;
        }
    }
    printf("Miller walk / G2 membership: %ld mismatches\n", bad3);

    // ---- line products on a quad ----
    long bad4 = 0;
    const HostQuad xq;
    for (int it = 0; it < 2000; it++) {
        const Fp12 f = rnd_fp12();
        const Fp2 a0 = rnd_fp2(), a1 = rnd_fp2(), a4 = rnd_fp2();
        Fp12Q<4> fq = to_quad(f);
        if (!eq(from_quad(fq), f)) { bad4++; continue; }
        LV<LineK::A0, 4> l0; LV<3, 4> l1, l4;
        for (int i = 0; i < 4; i++) {
            l0.l[i] = lift_big<LineK::A0>((i & 1) ? a0.c1 : a0.c0);
            l1.l[i] = lift_big<3>((i & 1) ? a1.c1 : a1.c0);
            l4.l[i] = lift_big<3>((i & 1) ? a4.c1 : a4.c0);
        }
        if (it & 1) quad_fold_line<true>(xq, fq, l0, l1, l4);          // both forms of the fold: six-product sums / nine two-product sums
        else quad_fold_line<false>(xq, fq, l0, l1, l4);
        if (!eq(from_quad(fq), mul_by_014(f, a0, a1, a4))) bad4++;
        const Fp12Q<4> seed = quad_seed_line(xq, l0, l1, l4);
        if (!eq(from_quad(seed), mul_by_014(fp12_one(), a0, a1, a4))) bad4++;
        // scaling of a stored record
        const Fp xs = rnd_fp();
        LV<LineK::A1D, 4> rx;
        for (int i = 0; i < 4; i++) rx.l[i] = lift_big<LineK::A1D>((i & 1) ? a1.c1 : a1.c0);
        const auto sd = line_scale_a1<false>(xq, rx, splat<4>(xs));
        if (!eq(join_q(sd, 0), mul_fp(sqr(a1), xs)) || !eq(join_q(sd, 1), mul_fp(sqr(a1), xs))) bad4++;
        LV<LineK::A1A, 4> rt;
        for (int i = 0; i < 4; i++) rt.l[i] = lift_big<LineK::A1A>((i & 1) ? a1.c1 : a1.c0);
        const auto sa = line_scale_a1<true>(xq, rt, splat<4>(xs));
        if (!eq(join_q(sa, 0), mul_fp(a1, xs))) bad4++;
    }
    printf("quad line products: %ld mismatches\n", bad4);

    // ---- dense products through memory ----
    long bad5 = 0;
    for (int it = 0; it < 300; it++) {
        const Fp12 f = rnd_fp12(), g = rnd_fp12();
        std::vector<uint32_t> ef(kElemWords), eg(kElemWords), eh(kElemWords);
        elem_from_fp12(ef.data(), f, 3);
        elem_from_fp12(eg.data(), g, 3);
        for (int k = 0; k < 6; k++)
            for (int q = 0; q < 2; q++) {
                FpL acc;
                if (it % 3 == 0) acc = dense_terms<6>(ef.data(), eg.data(), k, q, 0).l[0];
                else if (it % 3 == 1) acc = addL(dense_terms<3>(ef.data(), eg.data(), k, q, 0).l[0], dense_terms<3>(ef.data(), eg.data(), k, q, 3).l[0]);
                else {
                    acc = dense_terms<1>(ef.data(), eg.data(), k, q, 0).l[0];
                    for (int j = 1; j < 6; j++) acc = addL(acc, dense_terms<1>(ef.data(), eg.data(), k, q, j).l[0]);
                }
                elem_store(eh.data(), k, q, weak_reduceL(acc));
            }
        const Fp12 want = mul(f, g);
        const Fp2 w[6] = {want.c0.a0, want.c1.a0, want.c0.a1, want.c1.a1, want.c0.a2, want.c1.a2};
        for (int k = 0; k < 6; k++)
            if (!eq(canon_of(elem_load(eh.data(), k, 0)), w[k].c0) || !eq(canon_of(elem_load(eh.data(), k, 1)), w[k].c1)) bad5++;
    }
    printf("dense products: %ld mismatches\n", bad5);
    // ---- XYZZ points over Fp2 on 8 lanes (g2_limb.h: the G2 fold / bucket reduce) against curve.h ----
    long bad6 = 0;
    {
        auto to_k = [&](const Xyzz<Fp2> &a) {
            XyzzK<8> r;
            for (int i = 0; i < 8; i++) {
                const bool inf = is_zero(a.zz);
                r.x.l[i] = inf ? fpl_zero() : lift_big<8>((i & 1) ? a.x.c1 : a.x.c0);
                r.y.l[i] = inf ? fpl_zero() : lift_big<4>((i & 1) ? a.y.c1 : a.y.c0);
                r.zz.l[i] = inf ? fpl_zero() : lift_big<2>((i & 1) ? a.zz.c1 : a.zz.c0);
                r.zzz.l[i] = inf ? fpl_zero() : lift_big<2>((i & 1) ? a.zzz.c1 : a.zzz.c0);
            }
            return r;
        };
        auto same = [&](const XyzzK<8> &got, const Xyzz<Fp2> &want) {
            if (!lanes_agree(got.x) || !lanes_agree(got.y) || !lanes_agree(got.zz) || !lanes_agree(got.zzz)) return false;
            const bool ginf = is_inf8k(x8, got);
            if (ginf || is_zero(want.zz)) return ginf && is_zero(want.zz);
            const Aff<Fp2> a = to_affine(Xyzz<Fp2>{join_q(got.x), join_q(got.y), join_q(got.zz), join_q(got.zzz)}), b = to_affine(want);
            return eq(a.x, b.x) && eq(a.y, b.y);
        };
        auto lifted = [&](const Aff<Fp2> &a) {              // a random XYZZ representative of an affine point
            const Fp2 l = rnd_fp2(), ll = sqr(l), lll = mul(ll, l);
            return Xyzz<Fp2>{mul(a.x, ll), mul(a.y, lll), ll, lll};
        };
        const Xyzz<Fp2> inf = xyzz_inf<Fp2>();
        for (int it = 0; it < 200; it++) {
            const Aff<Fp2> A = (it % 5 == 4) ? map_to_curve<Fp2>(rnd_fp2()) : rnd_multiple(g2_gen()), B = rnd_multiple(g2_gen());
            const Xyzz<Fp2> a = lifted(A), b = lifted(B), a2 = lifted(A), na = Xyzz<Fp2>{a2.x, neg(a2.y), a2.zz, a2.zzz};
            if (!same(add8k(x8, to_k(a), to_k(b)), add(a, b))) bad6++;
            if (!same(add8k(x8, to_k(a), to_k(a2)), dbl(a))) bad6++;                 // equal points in different representatives
            if (!same(add8k(x8, to_k(a), to_k(na)), inf)) bad6++;                   // opposite points
            if (!same(add8k(x8, to_k(a), to_k(inf)), a) || !same(add8k(x8, to_k(inf), to_k(b)), b)) bad6++;
            if (!same(dbl8k(x8, to_k(a)), dbl(a)) || !same(dbl8k(x8, to_k(inf)), inf)) bad6++;
            const uint32_t m = it < 40 ? (uint32_t)it : (uint32_t)(rnd() % 5000);
            Xyzz<Fp2> want = inf;
            for (int bit = 12; bit >= 0; bit--) { want = dbl(want); if ((m >> bit) & 1u) want = add(want, a); }
            if (!same(small_mul8k(x8, to_k(a), m), want) || !same(small_mul8k(x8, to_k(inf), m), inf)) bad6++;
            // the running sums of a reduce segment: sum_{v = 1..4} v B_v with an empty bucket in the middle
            const Xyzz<Fp2> bk[4] = {a, inf, b, a2};
            XyzzK<8> R = xyzzk_inf<8>(), Q = xyzzk_inf<8>();
            Xyzz<Fp2> Rw = inf, Qw = inf;
            for (int v = 3; v >= 0; v--) { R = add8k(x8, R, to_k(bk[v])); Q = add8k(x8, Q, R); Rw = add(Rw, bk[v]); Qw = add(Qw, Rw); }
            if (!same(Q, Qw)) bad6++;
        }
    }
    printf("G2 XYZZ on 8 lanes: %ld mismatches\n", bad6);
    return (bad || bad1 || bad2 || bad3 || bad4 || bad5 || bad6) ? 1 : 0;
}
