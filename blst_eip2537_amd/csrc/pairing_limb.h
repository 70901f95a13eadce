// The pairing kernels' arithmetic in limb form (limbk.h): Miller line walk on the twist, G1 membership
// chain and the per-step line product, written once for a lane group and instantiated by pairing.hip
// (one share per lane, DPP exchanges) and by tools/pairing_limb_check.hip (all shares of a group in
// arrays, on the host, against pairing.h / curve.h).
//
// Replaces the arithmetic the reference obtains from blst inside bls12_pairing
// (src/eip2537.c:1033-1078: blst_p1_affine_in_g1, blst_p2_affine_in_g2, blst_miller_loop, blst_fp12_mul).
//
// Points are HOMOGENEOUS projective (X : Y : Z), x = X / Z, y = Y / Z, on y^2 = x^3 + b with b = 4 (G1)
// or 4 (1 + u) (twist).  The doubling of Renes-Costello-Batina (2016, algorithm 9, a = 0) has product
// depth TWO -- [X Y, Y^2, Z^2, Y Z] then [t0 XY, t0 (Y^2 + E), 8 Y^2 E, 8 Y^2 YZ] with E = 3 b Z^2,
// t0 = Y^2 - 3 E -- where the Jacobian / XYZZ forms of round 2 needed three rounds per step; a round of
// four products on the lanes of a group is the unit of time of these latency-bound kernels.  Both
// curves have odd order, so the RCB formulas are complete on them: the membership chains need no
// special cases for the small-order points that non-members are made of.
#pragma once
#include "limbk.h"

namespace eip {

// ---- how a group of lanes multiplies ---------------------------------------------------------------------
struct PolFp2c {        // Fp2 values split by component over 4 lane pairs (line walk)
    template <class X, class A0, class A1, class A2, class A3, class B0, class B1, class B2, class B3>
    static HDF auto round(const X &x, const A0 &a0, const A1 &a1, const A2 &a2, const A3 &a3, const B0 &b0, const B1 &b1,
                         const B2 &b2, const B3 &b3) {
        return round4_fp2(x, a0, a1, a2, a3, b0, b1, b2, b3);
    }
    template <class X, class A> static HDF auto twist(const X &x, const A &a) { return mul_xiB(x, a); }      // b / 4 = 1 + u
};
struct PolFp4 {         // whole Fp values replicated on 4 lanes (G1 membership)
    template <class X, class A0, class A1, class A2, class A3, class B0, class B1, class B2, class B3>
    static HDF auto round(const X &x, const A0 &a0, const A1 &a1, const A2 &a2, const A3 &a3, const B0 &b0, const B1 &b1,
                         const B2 &b2, const B3 &b3) {
        return round4_fp(x, a0, a1, a2, a3, b0, b1, b2, b3);
    }
    template <class X, class A> static HDF A twist(const X &, const A &a) { return a; }                     // b / 4 = 1
};

template <class A, class B, class C, class D, class E> struct Out5 { A x; B y; C z; D l0; E l1; };
template <class A, class B, class C, class D, class E> HDF Out5<A, B, C, D, E> out5(const A &a, const B &b, const C &c, const D &d, const E &e) {
    return Out5<A, B, C, D, E>{a, b, c, d, e};
}
template <class A, class B, class C, class D, class E, class F> struct Out6 { A x; B y; C z; D l0; E l1; F l2; };
template <class A, class B, class C, class D, class E, class F>
HDF Out6<A, B, C, D, E, F> out6(const A &a, const B &b, const C &c, const D &d, const E &e, const F &f) {
    return Out6<A, B, C, D, E, F>{a, b, c, d, e, f};
}

// 2 (X : Y : Z), two rounds.  Also returns what the tangent line at the point needs:
//   l0 = Y^2 - 3 b Z^2,  l1 = Y Z      (line: l0 + (-3 X^2) xP v + (2 Y Z) yP v w; X^2 is left to the product tree)
template <class Pol, class X, int KX, int KY, int KZ, int N>
HDF auto proj_dbl(const X &x, const LV<KX, N> &px, const LV<KY, N> &py, const LV<KZ, N> &pz) {
    const auto r1 = Pol::round(x, px, py, pz, pz, py, py, pz, py);             // X Y, Y^2, Z^2, Z Y
    const auto E = shlB<2>(mul3B(Pol::twist(x, r1.r2)));                       // 3 b Z^2 = 12 (twist) Z^2
    const auto t0 = subB(r1.r1, mul3B(E));                                     // Y^2 - 9 b Z^2
    const auto B8 = shlB<3>(r1.r1);
    const auto r2 = Pol::round(x, r1.r0, addB(r1.r1, E), E, r1.r3, t0, t0, B8, B8);
    return out5(dblB(r2.r0), addB(r2.r1, r2.r2), r2.r3, subB(r1.r1, E), r1.r3);
}

// ---- Miller walk on the twist ----------------------------------------------------------------------------
// Bounds (in units of p) of the running point between steps, and of the line records the walk stores
struct WalkK { static constexpr int X = 8, Y = 48, Z = 8; };
struct LineK {                       // doubling step: (Y^2 - E, X, Y Z);  addition step: (6 (theta xQ - lambda yQ), 2 theta, 3 lambda)
    static constexpr int A0 = 64, A1D = WalkK::X, A4D = 4, A1A = 104, A4A = 32;
};
template <int N> struct WalkPt { LV<WalkK::X, N> x; LV<WalkK::Y, N> y; LV<WalkK::Z, N> z; };
template <int N> struct LineRecD { LV<LineK::A0, N> a0; LV<LineK::A1D, N> a1; LV<LineK::A4D, N> a4; };
template <int N> struct LineRecA { LV<LineK::A0, N> a0; LV<LineK::A1A, N> a1; LV<LineK::A4A, N> a4; };

template <class X, int N> HDF LineRecD<N> miller_dbl_l(const X &x, WalkPt<N> &T) {
    const auto o = proj_dbl<PolFp2c>(x, T.x, T.y, T.z);
    LineRecD<N> l{widen<LineK::A0>(o.l0), T.x, widen<LineK::A4D>(o.l1)};
    T = WalkPt<N>{widen<WalkK::X>(o.x), widen<WalkK::Y>(o.y), widen<WalkK::Z>(o.z)};
    return l;
}
// T += Q (Q affine, canonical limbs), four rounds; line through T and Q scaled by 6 so that the product tree can
// use the same (-3 xP, 2 yP) factors as for a tangent:  6 (theta xQ - lambda yQ) + (2 theta)(-3 xP) v + (3 lambda)(2 yP) v w
// with theta = Y - yQ Z, lambda = X - xQ Z.  Not complete (T = +-Q gives Z = 0 from then on, which the
// membership test reads as "not in G2" -- such Q are not in G2).
template <class X, int N> HDF LineRecA<N> miller_add_l(const X &x, WalkPt<N> &T, const LV<1, N> &qx, const LV<1, N> &qy) {
    const auto r1 = PolFp2c::round(x, qy, qx, qy, qx, T.z, T.z, T.z, T.z);
    const auto theta = subB(T.y, r1.r0);
    const auto lambda = subB(T.x, r1.r1);
    const auto r2 = PolFp2c::round(x, theta, lambda, qx, qy, theta, lambda, theta, lambda);     // C = theta^2, D = lambda^2, xQ theta, yQ lambda
    const auto r3 = PolFp2c::round(x, lambda, T.z, T.x, T.x, r2.r1, r2.r0, r2.r1, r2.r1);       // E = lambda D, F = Z C, G = X D
    const auto H = subB(addB(r3.r0, r3.r1), dblB(r3.r2));                                         // E + F - 2 G
    const auto r4 = PolFp2c::round(x, lambda, theta, T.y, T.z, H, subB(r3.r2, H), r3.r0, r3.r0);
    LineRecA<N> l{widen<LineK::A0>(shlB<1>(mul3B(subB(r2.r2, r2.r3)))), widen<LineK::A1A>(dblB(theta)), widen<LineK::A4A>(mul3B(lambda))};
    T = WalkPt<N>{widen<WalkK::X>(r4.r0), widen<WalkK::Y>(subB(r4.r1, r4.r2)), widen<WalkK::Z>(r4.r3)};
    return l;
}
// After the walk T = [|z|] Q.  Q in G2  <=>  psi(Q) == [z] Q = -T:   psi(Q).x Z == X,  psi(Q).y Z == -Y,  Z != 0
// (blst_p2_affine_in_g2, reference src/eip2537.c:1051).  Per lane: true when Q is NOT a member.
template <class X, int N> HDF LanePred<N> g2_not_member_l(const X &x, const WalkPt<N> &T, const LV<1, N> &qx, const LV<1, N> &qy) {
    const auto cx = x.pick_q(qx, negB(qx)), cy = x.pick_q(qy, negB(qy));                    // conjugates: component 1 negated
    const auto kx = x.pick_q(lv_const<1, N>(FpL{{K_PSI_X_C0_R390_30}}), lv_const<1, N>(FpL{{K_PSI_X_C1_R390_30}}));
    const auto ky = x.pick_q(lv_const<1, N>(FpL{{K_PSI_Y_C0_R390_30}}), lv_const<1, N>(FpL{{K_PSI_Y_C1_R390_30}}));
    const auto r1 = PolFp2c::round(x, cx, cy, cx, cy, kx, ky, kx, ky);
    const auto r2 = PolFp2c::round(x, r1.r0, r1.r1, r1.r0, r1.r1, T.z, T.z, T.z, T.z);
    const auto okx = x.both(is_zero_modpB(subB(T.x, r2.r0)));
    const auto oky = x.both(is_zero_modpB(addB(T.y, r2.r1)));
    return !(okx & oky) | x.both(is_zero_modpB(T.z));
}

// ---- G1 membership ---------------------------------------------------------------------------------------
struct G1K { static constexpr int X = 8, Y = 16, Z = 8; };
template <int N> struct G1Pt { LV<G1K::X, N> x; LV<G1K::Y, N> y; LV<G1K::Z, N> z; };
// P1 + P2, complete (RCB algorithm 7, a = 0, 3 b = 12), three rounds of four products
template <class X, int N> HDF G1Pt<N> proj_add_g1(const X &x, const G1Pt<N> &a, const G1Pt<N> &b) {
    const auto r1 = round4_fp(x, a.x, a.y, a.z, addB(a.x, a.y), b.x, b.y, b.z, addB(b.x, b.y));         // t0, t1, t2, (X1 + Y1)(X2 + Y2)
    const auto t3 = subB(subB(r1.r3, r1.r0), r1.r1);                                                     // X1 Y2 + X2 Y1
    const auto t2b = shlB<2>(mul3B(r1.r2));                                                              // 3 b Z1 Z2
    const auto z3 = addB(r1.r1, t2b);
    const auto t1p = subB(r1.r1, t2b);
    const auto t0p = mul3B(r1.r0);
    const auto r2 = round4_fp(x, addB(a.y, a.z), addB(a.x, a.z), t3, t0p, addB(b.y, b.z), addB(b.x, b.z), t1p, t3);
    const auto t4 = subB(subB(r2.r0, r1.r1), r1.r2);
    const auto y3 = subB(subB(r2.r1, r1.r0), r1.r2);              // Y1 Z2 + Y2 Z1, X1 Z2 + X2 Z1
    const auto y3p = shlB<2>(mul3B(y3));
    const auto r3 = round4_fp(x, t4, t1p, t0p, t4, y3p, z3, y3p, z3);                                    // t4 y3', t1' z3, t0' y3', t4 z3
    return G1Pt<N>{widen<G1K::X>(subB(r2.r2, r3.r0)), widen<G1K::Y>(addB(r3.r2, r3.r1)), widen<G1K::Z>(addB(r3.r3, r2.r3))};
}
template <class X, int N> HDF G1Pt<N> g1_mul_zabs_l(const X &x, const G1Pt<N> &base) {
    const uint64_t z = K_Z_ABS;
    G1Pt<N> acc = base;
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
        const auto d = proj_dbl<PolFp4>(x, acc.x, acc.y, acc.z);
        acc = G1Pt<N>{widen<G1K::X>(d.x), widen<G1K::Y>(d.y), widen<G1K::Z>(d.z)};
        if ((z >> i) & 1ull) acc = proj_add_g1(x, acc, base);
    }
    return acc;
}
// phi(P) == -[z^2] P with phi(x, y) = (beta x, y)  (blst_p1_affine_in_g1, reference src/eip2537.c:1041).
// P affine, not infinity, canonical limbs.  Per lane: true when P is NOT a member.
template <class X, int N> HDF LanePred<N> g1_not_member_l(const X &x, const LV<1, N> &xp, const LV<1, N> &yp) {
    const G1Pt<N> P{widen<G1K::X>(xp), widen<G1K::Y>(yp), lv_const<G1K::Z, N>(fpl_one())};
    const G1Pt<N> T = g1_mul_zabs_l(x, g1_mul_zabs_l(x, P));
    const auto beta = lv_const<1, N>(FpL{{K_BETA_R390_30}});
    const auto r1 = round4_fp(x, xp, yp, xp, yp, beta, T.z, beta, T.z);                  // beta xP, yP Z
    const auto r2 = round4_fp(x, r1.r0, r1.r0, r1.r0, r1.r0, T.z, T.z, T.z, T.z);        // beta xP Z
    const auto okx = is_zero_modpB(subB(T.x, r2.r0));
    const auto oky = is_zero_modpB(addB(T.y, r1.r1));
    return !(okx & oky) | is_zero_modpB(T.z);
}

// ---- per-step line products: Fp12 on a quad of lanes --------------------------------------------------------
// f = sum_k f_k w^k (f_k in Fp2, w^6 = 1 + u).  Lane (c, q) of a quad holds component q of three coefficients:
//   c = 0:  own[t] = f_{2t}              (f0, f2, f4)
//   c = 1:  own[t] = f_{2((t+1) mod 3)+1} (f3, f5, f1)   -- rotated by one, so that in  f * (a0 + a1 w^2 + a4 w^3)
//   new own[t] = own[t] a0 + own[t-1] A1[t] + other[t] A4[t]
// reads the same register index on both halves; only the line operand differs (xi = 1 + u folded into it):
//   c = 0:  A1 = (xi a1, a1, a1),  A4 = (xi a4, xi a4, a4)         c = 1:  A1 = (a1, a1, xi a1),  A4 = (a4, a4, xi a4)
// A product f_j * L seen from component q is  f_j.0 A + f_j.1 B  with  A = L.q,  B = q ? L.0 : -L.1.
struct TreeK { static constexpr int F = 64; };
template <int N> struct Fp12Q { LV<TreeK::F, N> own[3]; };

// scaled line (a1 = X^2 xs or (2 theta) xs, a4 = (l1) ys) from a stored record; xs = -3 xP, ys = 2 yP (Fp: the same on every lane)
template <bool ADD, class X, int K1, int N>
HDF auto line_scale_a1(const X &x, const LV<K1, N> &r1, const LV<1, N> &xs) {
    if constexpr (ADD) {
        return mulB(r1, xs);
    } else {
        const auto rp = x.swap(r1);                       // X^2: component 0 = (x0 + x1)(x0 - x1), component 1 = 2 x0 x1
        const auto sq = mulB(x.pick_q(addB(r1, rp), dblB(r1)), x.pick_q(subB(r1, rp), rp));
        return mulB(sq, xs);
    }
}
// f <- f * line.  a0 / a1 / a4: this lane's component of the line's coefficients (a1, a4 already scaled).
// Term-major: the three products by a0 first, then those by a1 / xi a1, then a4 / xi a4, so that only one
// coefficient's operand forms (A = own component, B = q ? partner : -partner) are alive at a time.
template <class X, int KL, int N> struct QuadOperand { LV<KL, N> A, B; };
template <class X, int KL, int N> HDF QuadOperand<X, KL, N> quad_operand(const X &x, const LV<KL, N> &l) {
    const auto lp = x.swap(l);
    return QuadOperand<X, KL, N>{l, x.pick_q(negB(lp), lp)};
}
// MUL6 (round 4): coefficient-major, the three two-product sums of a coefficient in ONE six-product sum with a single reduction (mul6B):
// 3 x 1 183 multiply-adds per lane and line instead of 9 x 507, but the operand forms of all three line coefficients are alive at once --
// 349 registers: the kernel that takes this form runs one wave per SIMD (k_pair_fold<false, true>, checks of up to 2^13 pairs).
template <bool MUL6, class X, int K0, int K1, int K4, int N>
HDF void quad_fold_line(const X &x, Fp12Q<N> &f, const LV<K0, N> &a0, const LV<K1, N> &a1, const LV<K4, N> &a4) {
    constexpr int K1X = 2 * K1, K4X = 2 * K4;
    if constexpr (MUL6) {
    const auto o0 = quad_operand(x, a0);
    const auto o1 = quad_operand(x, widen<K1X>(a1));
    const auto o1x = quad_operand(x, addB(a1, x.pick_q(negB(x.swap(a1)), x.swap(a1))));      // component q of (1 + u) a1
    const auto o4 = quad_operand(x, widen<K4X>(a4));
    const auto o4x = quad_operand(x, addB(a4, x.pick_q(negB(x.swap(a4)), x.swap(a4))));
    const auto A01 = x.pick_c(o4x.A, o4.A), B01 = x.pick_c(o4x.B, o4.B);
    const auto n0 = mul6B(x.template same_c<0>(f.own[0]), o0.A, x.template same_c<1>(f.own[0]), o0.B,
                          x.template same_c<0>(f.own[2]), x.pick_c(o1x.A, o1.A), x.template same_c<1>(f.own[2]), x.pick_c(o1x.B, o1.B),
                          x.template other_c<0>(f.own[0]), A01, x.template other_c<1>(f.own[0]), B01);
    const auto n1 = mul6B(x.template same_c<0>(f.own[1]), o0.A, x.template same_c<1>(f.own[1]), o0.B,
                          x.template same_c<0>(f.own[0]), o1.A, x.template same_c<1>(f.own[0]), o1.B,
                          x.template other_c<0>(f.own[1]), A01, x.template other_c<1>(f.own[1]), B01);
    const auto n2 = mul6B(x.template same_c<0>(f.own[2]), o0.A, x.template same_c<1>(f.own[2]), o0.B,
                          x.template same_c<0>(f.own[1]), x.pick_c(o1.A, o1x.A), x.template same_c<1>(f.own[1]), x.pick_c(o1.B, o1x.B),
                          x.template other_c<0>(f.own[2]), x.pick_c(o4.A, o4x.A), x.template other_c<1>(f.own[2]), x.pick_c(o4.B, o4x.B));
    f.own[0] = widen<TreeK::F>(n0);
    f.own[1] = widen<TreeK::F>(n1);
    f.own[2] = widen<TreeK::F>(n2);
    } else {
    // term-major: the three products by a0 first, then those by a1 / xi a1, then a4 / xi a4, so that only one coefficient's operand
    // forms are alive at a time (256 registers at two waves per SIMD)
    // own[t] a0
    const auto o0 = quad_operand(x, a0);
    const auto p0 = mul2B(x.template same_c<0>(f.own[0]), o0.A, x.template same_c<1>(f.own[0]), o0.B);
    const auto p1 = mul2B(x.template same_c<0>(f.own[1]), o0.A, x.template same_c<1>(f.own[1]), o0.B);
    const auto p2 = mul2B(x.template same_c<0>(f.own[2]), o0.A, x.template same_c<1>(f.own[2]), o0.B);
    // own[t - 1] A1[t]:   c = 0: (xi a1, a1, a1)    c = 1: (a1, a1, xi a1)
    const auto o1 = quad_operand(x, widen<K1X>(a1));
    const auto o1x = quad_operand(x, addB(a1, x.pick_q(negB(x.swap(a1)), x.swap(a1))));      // component q of (1 + u) a1
    const auto q1 = addB(p1, mul2B(x.template same_c<0>(f.own[0]), o1.A, x.template same_c<1>(f.own[0]), o1.B));
    const auto q0 = addB(p0, mul2B(x.template same_c<0>(f.own[2]), x.pick_c(o1x.A, o1.A), x.template same_c<1>(f.own[2]), x.pick_c(o1x.B, o1.B)));
    const auto q2 = addB(p2, mul2B(x.template same_c<0>(f.own[1]), x.pick_c(o1.A, o1x.A), x.template same_c<1>(f.own[1]), x.pick_c(o1.B, o1x.B)));
    // other[t] A4[t]:     c = 0: (xi a4, xi a4, a4)    c = 1: (a4, a4, xi a4)
    const auto o4 = quad_operand(x, widen<K4X>(a4));
    const auto o4x = quad_operand(x, addB(a4, x.pick_q(negB(x.swap(a4)), x.swap(a4))));
    const auto A01 = x.pick_c(o4x.A, o4.A), B01 = x.pick_c(o4x.B, o4.B);
    const auto n0 = addB(q0, mul2B(x.template other_c<0>(f.own[0]), A01, x.template other_c<1>(f.own[0]), B01));
    const auto n1 = addB(q1, mul2B(x.template other_c<0>(f.own[1]), A01, x.template other_c<1>(f.own[1]), B01));
    const auto n2 = addB(q2, mul2B(x.template other_c<0>(f.own[2]), x.pick_c(o4.A, o4x.A), x.template other_c<1>(f.own[2]), x.pick_c(o4.B, o4x.B)));
    f.own[0] = widen<TreeK::F>(n0);
    f.own[1] = widen<TreeK::F>(n1);
    f.own[2] = widen<TreeK::F>(n2);
    }
}
// f = the line itself (first line of a quad): slots w^0, w^2, w^3
template <class X, int K0, int K1, int K4, int N>
HDF Fp12Q<N> quad_seed_line(const X &x, const LV<K0, N> &a0, const LV<K1, N> &a1, const LV<K4, N> &a4) {
    const auto z = lv_const<TreeK::F, N>(fpl_zero());
    Fp12Q<N> f;
    f.own[0] = x.pick_c(widen<TreeK::F>(a0), widen<TreeK::F>(a4));       // c = 0: f0 = a0      c = 1: f3 = a4
    f.own[1] = x.pick_c(widen<TreeK::F>(a1), z);                          // c = 0: f2 = a1      c = 1: f5 = 0
    f.own[2] = z;                                                         //        f4 = 0              f1 = 0
    return f;
}

// ---- dense products through shared memory --------------------------------------------------------------------
// An Fp12 element in memory: 12 limb strings [k][q] (coefficient of w^k, component q), kElemStride dwords each.
static constexpr int kLimbStride = 14, kElemWords = 12 * kLimbStride;
HDF FpL elem_load(const uint32_t *e, int k, int q) {
    FpL v;
    const uint32_t *s = e + (k * 2 + q) * kLimbStride;
#pragma unroll
    for (int i = 0; i < 13; i++) v.l[i] = s[i];
    return v;
}
HDF void elem_store(uint32_t *e, int k, int q, const FpL &v) {
    uint32_t *d = e + (k * 2 + q) * kLimbStride;
#pragma unroll
    for (int i = 0; i < 13; i++) d[i] = v.l[i];
}
// UPL consecutive terms j0 .. j0 + UPL - 1 of component q of coefficient k of f * g  (operands at most 3 p):
//   sum_j (f_j * g_{k-j} * (xi if j > k)).q        each term one two-product sum
template <int UPL> HDF LV<2 * UPL + 1, 1> dense_terms(const uint32_t *f, const uint32_t *g, int k, int q, int j0) {
    using V = LV<3, 1>;
    LV<2 * UPL + 1, 1> acc;
#pragma unroll
    for (int u = 0; u < UPL; u++) {
        const int j = j0 + u, kk = (k - j + 6) % 6;
        const bool wrap = j > k;
        const V f0{{elem_load(f, j, 0)}}, f1{{elem_load(f, j, 1)}}, g0{{elem_load(g, kk, 0)}}, g1{{elem_load(g, kk, 1)}};
        // G = g_kk or (1 + u) g_kk = (g0 - g1, g0 + g1);   A = G.q,  B = q ? G.0 : -G.1
        const auto d = subB(g0, g1);
        const auto s = addB(g0, g1);                                            // <= 6
        // selections by value, limb by limb (an lvalue conditional on whole values selects an ADDRESS and pins both in scratch)
        const auto nG1w = negB(s), nG1 = negB(widen<6>(g1));
        LV<6, 1> A, B;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const uint32_t G0 = pickv(wrap, d.l[0].l[i], g0.l[0].l[i]), G1 = pickv(wrap, s.l[0].l[i], g1.l[0].l[i]);
            A.l[0].l[i] = pickv(q != 0, G1, G0);
            B.l[0].l[i] = pickv(q != 0, G0, pickv(wrap, nG1w.l[0].l[i], nG1.l[0].l[i]));
        }
        const auto t = mul2B(f0, A, f1, B);                                     // 2 * 3 * 6 / 630 + 2 = 2
        static_assert(decltype(t)::kK == 2, "");
        if (u == 0) acc.l[0] = t.l[0];
        else acc.l[0] = addL(acc.l[0], t.l[0]);
    }
    return acc;
}

}  // namespace eip
