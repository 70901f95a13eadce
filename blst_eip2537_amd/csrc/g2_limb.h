// XYZZ points over Fp2 on 8-lane groups, in limb form with compile-time bounds (limbk.h) -- the point operations of the G2
// fold / bucket-reduce kernels (msm.hip: k_msm_fold_small8c_l, k_msm_fold_big8c_l, k_msm_reduce8c_l, k_msm_window_sum8c_l).
//
// lanes.h's add8c / dbl8c do the same rounds on FpI words (12 x 32 bits, re-sliced into 30-bit limbs around every product,
// conditional corrections in every linear step); here a value stays in 13 limbs from the accumulate (k_msm_accum2c_l) to the
// window sums, differences are a + K p - b and every bound is part of the type.  A group of 8 lanes = 4 lane pairs; lane (p, q)
// holds component q of every coordinate (replicated over the pairs); a round multiplies four pairs of Fp2 operands, lane pair p
// computing product p (round4_fp2, limbk.h).  Written over an executor X (device: DevLanes8, dev_lanes.h; host: HostLanes8), so
// tools/pairing_limb_check.hip runs the SAME code on the host against curve.h.
//
// Bounds of a stored point, in units of p: x <= 8, y <= 4, zz, zzz <= 2 (what the accumulate leaves and what every operation here
// returns).  The point at infinity is zz = 0 in all limbs of both components: a computed zz is a product of Fp2 values that are not
// 0, so one of its components is not a multiple of p, and a product with an all-zero FIRST operand is exactly 0 limbs (the second
// operand's component enters as B p - v, which is not) -- so zz and zzz always go on the a side of a round.
#pragma once
#include "limbk.h"

namespace eip {

template <int N> struct XyzzK { LV<8, N> x; LV<4, N> y; LV<2, N> zz; LV<2, N> zzz; };

template <int K, int N> HDF LanePred<N> is_zeroB(const LV<K, N> &a) {       // all limbs 0 (the infinity marker)
    LanePred<N> r;
    EIP_EACH_LANE r.b[i] = is_zero(a.l[i]);
    return r;
}
template <int N> HDF XyzzK<N> xyzzk_inf() {
    XyzzK<N> r;
    EIP_EACH_LANE { r.x.l[i] = fpl_zero(); r.y.l[i] = fpl_zero(); r.zz.l[i] = fpl_zero(); r.zzz.l[i] = fpl_zero(); }
    return r;
}
// uniform in the group: every lane of the group sees the same points
template <class X, int N> HDF bool is_inf8k(const X &x, const XyzzK<N> &p) { return x.both(is_zeroB(p.zz)).b[0]; }

// 2P (dbl-2008-s-1), three rounds; infinity stays infinity
template <class X, int N> HDF XyzzK<N> dbl8k(const X &x, const XyzzK<N> &p) {
    const auto U = dblB(p.y);                                                   // 8
    const auto r1 = round4_fp2(x, U, p.x, U, p.x, U, p.x, U, p.x);
    const auto &V = r1.r0, &XX = r1.r1;                                         // 2, 2
    const auto M = mul3B(XX);                                                   // 6
    const auto r2 = round4_fp2(x, U, p.x, M, p.zz, V, V, M, V);                 // zz on the a side
    const auto &W = r2.r0, &S = r2.r1, &MM = r2.r2, &ZZ3 = r2.r3;
    const auto X3 = subB(MM, dblB(S));                                          // 6
    const auto r3 = round4_fp2(x, M, W, W, W, subB(S, X3), p.y, p.zzz, p.zzz);  // W is exactly 0 when p is infinity
    XyzzK<N> o;
    o.x = widen<8>(X3);
    o.y = subB(r3.r0, r3.r1);
    o.zz = widen<2>(ZZ3);
    o.zzz = widen<2>(r3.r2);
    return o;
}
// P + Q (add-2008-s), four rounds, complete
template <class X, int N> HDF XyzzK<N> add8k(const X &x, const XyzzK<N> &p, const XyzzK<N> &q) {
    if (is_inf8k(x, q)) return p;
    if (is_inf8k(x, p)) return q;
    const auto r1 = round4_fp2(x, p.x, q.x, p.y, q.y, q.zz, p.zz, q.zzz, p.zzz);
    const auto &U1 = r1.r0, &U2 = r1.r1, &S1 = r1.r2, &S2 = r1.r3;              // 2 each
    const auto Pd = subB(U2, U1), Rr = subB(S2, S1);                            // 4, 4
    if (x.both(is_zero_modpB(Pd)).b[0]) {                                       // same x: double or cancel (rare)
        if (x.both(is_zero_modpB(Rr)).b[0]) return dbl8k(x, p);
        return xyzzk_inf<N>();
    }
    const auto r2 = round4_fp2(x, Pd, Rr, p.zz, p.zzz, Pd, Rr, q.zz, q.zzz);
    const auto &PP = r2.r0, &RR = r2.r1, &ZZ12 = r2.r2, &ZZZ12 = r2.r3;
    const auto r3 = round4_fp2(x, Pd, U1, ZZ12, ZZ12, PP, PP, PP, PP);
    const auto &PPP = r3.r0, &Q = r3.r1, &ZZ3 = r3.r2;
    const auto X3 = subB(subB(RR, PPP), dblB(Q));                               // 8
    const auto r4 = round4_fp2(x, Rr, S1, ZZZ12, ZZZ12, subB(Q, X3), PPP, PPP, PPP);
    XyzzK<N> o;
    o.x = widen<8>(X3);
    o.y = subB(r4.r0, r4.r1);
    o.zz = widen<2>(ZZ3);
    o.zzz = widen<2>(r4.r2);
    return o;
}
// m P by double-and-add from the top bit
template <class X, int N> HDF XyzzK<N> small_mul8k(const X &x, const XyzzK<N> &p, uint32_t m) {
    if (m == 0) return xyzzk_inf<N>();
    int top = 31;
    while (!((m >> top) & 1u)) top--;
    XyzzK<N> acc = p;
    for (int i = top - 1; i >= 0; i--) {
        acc = dbl8k(x, acc);
        if ((m >> i) & 1u) acc = add8k(x, acc, p);
    }
    return acc;
}

}  // namespace eip
