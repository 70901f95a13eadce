// Limb-form values with COMPILE-TIME bounds, written once for "all lanes of a lane group".
//
// limb30.h keeps Fp values as 13 limbs of 30 bits with Montgomery factor R' = 2^390 and tracks the size
// of every value by hand in comments.  The pairing kernels (pairing.hip: line walk, G1 membership,
// per-step product tree) have far more intermediate values than the G1 accumulate, so here the bound
// is part of the type:
//     LV<K, N>   the shares that the N lanes of a lane group hold of one value; every share is an FpL
//                with  0 <= value <= K p,  limbs 0 .. 11 below 2^30.
// Sums add bounds, a difference a - b is a + B p - b (bound A + B), a Montgomery product of bounds A, B
// is at most (A B / 630 + 1) p because R' / p = 630.05 -- so products contract and a loop is sound when
// the bounds of the values it carries do not grow, which static_assert checks where the loop closes.
// Operands of a product must stay <= kMaxK p so that their top limb is below 2^30 (column bounds of
// mulL / mul2L in limb30.h).
//
// N: the device kernels instantiate everything with N = 1 (a lane holds its own share, exchanges are DPP
// moves inside the executor types of pairing.hip); tools/pairing_limb_check.hip instantiates the SAME
// code with N = 4 / 8 and array permutations for the exchanges, and checks it on the host against
// pairing.h / curve.h.
#pragma once
#include "limb30.h"

namespace eip {

// everything here is inlined into the kernel that uses it (a call would pass limb strings through the stack)
#define HDF __host__ __device__ __forceinline__

static constexpr int kMaxK = 600;      // 600 p / 2^360 < 2^30
static constexpr int kMaxSum = 2000;   // any value: top limb below 2^32 with room

template <int K, int N> struct LV {
    static_assert(K >= 1 && K <= kMaxSum, "limb-form bound out of range");
    static constexpr int kK = K;
    FpL l[N];
};
template <int N> struct LanePred { bool b[N]; };

constexpr int prod_bound(long ab) { return (int)(ab / 630) + 2; }
constexpr int max2(int a, int b) { return a > b ? a : b; }
constexpr int max4(int a, int b, int c, int d) { return max2(max2(a, b), max2(c, d)); }

#define EIP_EACH_LANE _Pragma("unroll") for (int i = 0; i < N; i++)

// selection BY VALUE: "c ? a.l[i] : b.l[i]" on lvalues selects an address and pins both values in scratch memory
HDF uint32_t pickv(bool c, uint32_t a, uint32_t b) { return c ? a : b; }

template <int K2, int K, int N> HDF LV<K2, N> widen(const LV<K, N> &a) {
    static_assert(K2 >= K, "widen() cannot shrink a bound");
    LV<K2, N> r;
    EIP_EACH_LANE r.l[i] = a.l[i];
    return r;
}
template <int A, int B, int N> HDF LV<A + B, N> addB(const LV<A, N> &a, const LV<B, N> &b) {
    LV<A + B, N> r;
    EIP_EACH_LANE r.l[i] = addL(a.l[i], b.l[i]);
    return r;
}
// a - b as a + B p - b
template <int A, int B, int N> HDF LV<A + B, N> subB(const LV<A, N> &a, const LV<B, N> &b) {
    LV<A + B, N> r;
    EIP_EACH_LANE r.l[i] = subL<B>(a.l[i], b.l[i]);
    return r;
}
template <int B, int N> HDF LV<B, N> negB(const LV<B, N> &b) {
    LV<B, N> r;
    EIP_EACH_LANE r.l[i] = negL<B>(b.l[i]);
    return r;
}
template <int A, int N> HDF LV<2 * A, N> dblB(const LV<A, N> &a) {
    LV<2 * A, N> r;
    EIP_EACH_LANE r.l[i] = addL(a.l[i], a.l[i]);
    return r;
}
template <int A, int N> HDF LV<3 * A, N> mul3B(const LV<A, N> &a) {
    LV<3 * A, N> r;
    EIP_EACH_LANE r.l[i] = dbl_addL(a.l[i], a.l[i]);
    return r;
}
// a * 2^S: a shift of the limb string, no carries (the two parts of a limb do not overlap)
template <int S> HDF FpL shlL(const FpL &a) {
    static_assert(S >= 1 && S <= 4, "");
    FpL r;
#pragma unroll
    for (int k = 0; k < 12; k++) r.l[k] = ((a.l[k] << S) & kM30) | (k ? a.l[k - 1] >> (30 - S) : 0u);
    r.l[12] = (a.l[12] << S) | (a.l[11] >> (30 - S));
    return r;
}
template <int S, int A, int N> HDF LV<(A << S), N> shlB(const LV<A, N> &a) {
    LV<(A << S), N> r;
    EIP_EACH_LANE r.l[i] = shlL<S>(a.l[i]);
    return r;
}
// Weak reduction: a value of up to K p (K <= kMaxK) comes back congruent and at most 3 p.  With t the top limb
// (the value / 2^360, below 2^30) and ph = floor(p / 2^360) + 1, q = floor(t M / 2^52) with M = floor(2^52 / ph)
// never exceeds t / ph, so q p <= value, and it is short of value / p by less than 2: the rest is below 3 p.
// One multiplication for q, then value - q p in one signed carry pass (q p30[k] < 2^40).
HDF FpL weak_reduceL(const FpL &a) {
    const uint32_t p30[13] = {K_P30};
    constexpr uint64_t ph = (uint64_t)0x001a0111u + 1u;
    constexpr uint32_t M = (uint32_t)((1ull << 52) / ph);
    const uint32_t q = (uint32_t)(((uint64_t)a.l[12] * M) >> 52);
    FpL r;
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int64_t t = (int64_t)a.l[k] - (int64_t)((uint64_t)q * p30[k]) + c;
        r.l[k] = (uint32_t)t & kM30;
        c = t >> 30;
    }
    r.l[12] = (uint32_t)((int64_t)a.l[12] - (int64_t)((uint64_t)q * p30[12]) + c);
    return r;
}
template <int A, int N> HDF LV<3, N> weak_reduceB(const LV<A, N> &a) {
    static_assert(A <= kMaxK, "weak reduction needs a top limb below 2^30");
    LV<3, N> r;
    EIP_EACH_LANE r.l[i] = weak_reduceL(a.l[i]);
    return r;
}
template <int A, int B, int N> HDF LV<prod_bound((long)A * B), N> mulB(const LV<A, N> &a, const LV<B, N> &b) {
    static_assert(A <= kMaxK && B <= kMaxK, "product operand too large");
    LV<prod_bound((long)A * B), N> r;
    EIP_EACH_LANE r.l[i] = mulL(a.l[i], b.l[i]);
    return r;
}
template <int A, int N> HDF LV<prod_bound((long)A * A), N> sqrB(const LV<A, N> &a) {
    static_assert(A <= kMaxK, "product operand too large");
    LV<prod_bound((long)A * A), N> r;
    EIP_EACH_LANE r.l[i] = sqrL(a.l[i]);
    return r;
}
// a b + c d with one reduction
template <int A, int B, int C, int D, int N>
HDF LV<prod_bound((long)A * B + (long)C * D), N> mul2B(const LV<A, N> &a, const LV<B, N> &b, const LV<C, N> &c, const LV<D, N> &d) {
    static_assert(A <= kMaxK && B <= kMaxK && C <= kMaxK && D <= kMaxK, "product operand too large");
    LV<prod_bound((long)A * B + (long)C * D), N> r;
    EIP_EACH_LANE r.l[i] = mul2L(a.l[i], b.l[i], c.l[i], d.l[i]);
    return r;
}
// a0 b0 + ... + a5 b5 with one reduction (limb30.h: mul6L)
template <int A0, int B0, int A1, int B1, int A2, int B2, int A3, int B3, int A4, int B4, int A5, int B5, int N>
HDF LV<prod_bound((long)A0 * B0 + (long)A1 * B1 + (long)A2 * B2 + (long)A3 * B3 + (long)A4 * B4 + (long)A5 * B5), N>
mul6B(const LV<A0, N> &a0, const LV<B0, N> &b0, const LV<A1, N> &a1, const LV<B1, N> &b1, const LV<A2, N> &a2, const LV<B2, N> &b2,
      const LV<A3, N> &a3, const LV<B3, N> &b3, const LV<A4, N> &a4, const LV<B4, N> &b4, const LV<A5, N> &a5, const LV<B5, N> &b5) {
    static_assert(A0 <= kMaxK && B0 <= kMaxK && A1 <= kMaxK && B1 <= kMaxK && A2 <= kMaxK && B2 <= kMaxK && A3 <= kMaxK && B3 <= kMaxK &&
                  A4 <= kMaxK && B4 <= kMaxK && A5 <= kMaxK && B5 <= kMaxK, "product operand too large");
    LV<prod_bound((long)A0 * B0 + (long)A1 * B1 + (long)A2 * B2 + (long)A3 * B3 + (long)A4 * B4 + (long)A5 * B5), N> r;
    EIP_EACH_LANE r.l[i] = mul6L(a0.l[i], b0.l[i], a1.l[i], b1.l[i], a2.l[i], b2.l[i], a3.l[i], b3.l[i], a4.l[i], b4.l[i], a5.l[i], b5.l[i]);
    return r;
}
// value == 0 mod p, per lane
template <int K, int N> HDF LanePred<N> is_zero_modpB(const LV<K, N> &a) {
    LanePred<N> r;
    EIP_EACH_LANE r.b[i] = is_zero_modp(a.l[i], (uint32_t)K + 1u);
    return r;
}
template <int N> HDF LanePred<N> operator&(const LanePred<N> &a, const LanePred<N> &b) {
    LanePred<N> r;
    EIP_EACH_LANE r.b[i] = a.b[i] && b.b[i];
    return r;
}
template <int N> HDF LanePred<N> operator|(const LanePred<N> &a, const LanePred<N> &b) {
    LanePred<N> r;
    EIP_EACH_LANE r.b[i] = a.b[i] || b.b[i];
    return r;
}
template <int N> HDF LanePred<N> operator!(const LanePred<N> &a) {
    LanePred<N> r;
    EIP_EACH_LANE r.b[i] = !a.b[i];
    return r;
}
template <int K, int N> HDF LV<K, N> lv_const(const FpL &c) {
    LV<K, N> r;
    EIP_EACH_LANE r.l[i] = c;
    return r;
}

// ---- executors for the host emulation (the device ones live in pairing.hip) ---------------------------
// Group of 8 lanes = 4 lane pairs; lane (p, q) = index 2 p + q holds component q of every Fp2 value.
struct HostLanes8 {
    static constexpr int N = 8;
    template <int K> LV<K, 8> swap(const LV<K, 8> &a) const {
        LV<K, 8> r;
        for (int i = 0; i < 8; i++) r.l[i] = a.l[i ^ 1];
        return r;
    }
    template <int J, int K> LV<K, 8> from_pair(const LV<K, 8> &a) const {
        LV<K, 8> r;
        for (int i = 0; i < 8; i++) r.l[i] = a.l[2 * J + (i & 1)];
        return r;
    }
    template <int A, int B> LV<max2(A, B), 8> pick_q(const LV<A, 8> &a, const LV<B, 8> &b) const {      // q == 0 ? a : b
        LV<max2(A, B), 8> r;
        for (int i = 0; i < 8; i++) r.l[i] = (i & 1) == 0 ? a.l[i] : b.l[i];
        return r;
    }
    template <int A, int B, int C, int D>
    LV<max4(A, B, C, D), 8> pick_p(const LV<A, 8> &a, const LV<B, 8> &b, const LV<C, 8> &c, const LV<D, 8> &d) const {
        LV<max4(A, B, C, D), 8> r;
        for (int i = 0; i < 8; i++) r.l[i] = (i >> 1) == 0 ? a.l[i] : (i >> 1) == 1 ? b.l[i] : (i >> 1) == 2 ? c.l[i] : d.l[i];
        return r;
    }
    LanePred<8> both(const LanePred<8> &m) const {                   // the predicate holds on both components
        LanePred<8> r;
        for (int i = 0; i < 8; i++) r.b[i] = m.b[i] && m.b[i ^ 1];
        return r;
    }
};
// Group of 4 lanes, every lane holds whole Fp values (replicated); lane r computes product r of a round.
struct HostLanes4 {
    static constexpr int N = 4;
    template <int J, int K> LV<K, 4> from_lane(const LV<K, 4> &a) const {
        LV<K, 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = a.l[J];
        return r;
    }
    template <int A, int B, int C, int D>
    LV<max4(A, B, C, D), 4> pick_r(const LV<A, 4> &a, const LV<B, 4> &b, const LV<C, 4> &c, const LV<D, 4> &d) const {
        LV<max4(A, B, C, D), 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = i == 0 ? a.l[i] : i == 1 ? b.l[i] : i == 2 ? c.l[i] : d.l[i];
        return r;
    }
};

// Group of 4 lanes of the product tree: lane (c, q) = index 2 c + q holds component q of three of the six Fp2
// coefficients of an Fp12 value (c = 0: the even powers of w, c = 1: the odd ones).
struct HostQuad {
    static constexpr int N = 4;
    template <int K> LV<K, 4> swap(const LV<K, 4> &a) const {            // same c, other component
        LV<K, 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = a.l[i ^ 1];
        return r;
    }
    template <int Q, int K> LV<K, 4> same_c(const LV<K, 4> &a) const {   // component Q of the own half
        LV<K, 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = a.l[(i & 2) + Q];
        return r;
    }
    template <int Q, int K> LV<K, 4> other_c(const LV<K, 4> &a) const {  // component Q of the other half
        LV<K, 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = a.l[((i & 2) ^ 2) + Q];
        return r;
    }
    template <int A, int B> LV<max2(A, B), 4> pick_q(const LV<A, 4> &a, const LV<B, 4> &b) const {
        LV<max2(A, B), 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = (i & 1) == 0 ? a.l[i] : b.l[i];
        return r;
    }
    template <int A, int B> LV<max2(A, B), 4> pick_c(const LV<A, 4> &a, const LV<B, 4> &b) const {      // c == 0 ? a : b
        LV<max2(A, B), 4> r;
        for (int i = 0; i < 4; i++) r.l[i] = (i & 2) == 0 ? a.l[i] : b.l[i];
        return r;
    }
};

// ---- rounds of four products --------------------------------------------------------------------------
template <int K0, int K1, int K2, int K3, int N> struct Prod4 { LV<K0, N> r0; LV<K1, N> r1; LV<K2, N> r2; LV<K3, N> r3; };

// Fp2 values split by component over lane pairs: lane pair j multiplies (a_j, b_j) by the schoolbook rule
// (component 0: a0 b0 - a1 b1, component 1: a0 b1 + a1 b0 -- one two-product sum with a single reduction per
// lane) and every lane gets its component of all four results.
// Bound of a slot's result: the difference of component 0 is taken as  up (BM p - vp)  with BM the largest
// bound among the four second operands, so a slot's sum is at most A_j (B_j + BM) p^2: put the operand whose
// bounds are large and alike on the b side.
constexpr int pb2(int a, int b, int bm) { return prod_bound((long)a * (b + bm)); }
template <class X, int A0, int A1, int A2, int A3, int B0, int B1, int B2, int B3, int N>
HDF Prod4<pb2(A0, B0, max4(B0, B1, B2, B3)), pb2(A1, B1, max4(B0, B1, B2, B3)), pb2(A2, B2, max4(B0, B1, B2, B3)), pb2(A3, B3, max4(B0, B1, B2, B3)), N>
round4_fp2(const X &x, const LV<A0, N> &a0, const LV<A1, N> &a1, const LV<A2, N> &a2, const LV<A3, N> &a3,
           const LV<B0, N> &b0, const LV<B1, N> &b1, const LV<B2, N> &b2, const LV<B3, N> &b3) {
    static_assert(max4(A0, A1, A2, A3) <= kMaxK && max4(B0, B1, B2, B3) <= kMaxK, "product operand too large");
    constexpr int BM = max4(B0, B1, B2, B3);
    const auto u = x.pick_p(a0, a1, a2, a3);
    const auto v = x.pick_p(b0, b1, b2, b3);
    const auto up = x.swap(u);
    const auto vp = x.swap(v);
    // q = 0:  u v + up (BM p - vp)          q = 1:  up v + u vp
    const auto c = mul2B(x.pick_q(u, up), v, x.pick_q(up, u), x.pick_q(negB(vp), vp));
    Prod4<pb2(A0, B0, BM), pb2(A1, B1, BM), pb2(A2, B2, BM), pb2(A3, B3, BM), N> r;
    const auto c0 = x.template from_pair<0>(c), c1 = x.template from_pair<1>(c), c2 = x.template from_pair<2>(c), c3 = x.template from_pair<3>(c);
    EIP_EACH_LANE { r.r0.l[i] = c0.l[i]; r.r1.l[i] = c1.l[i]; r.r2.l[i] = c2.l[i]; r.r3.l[i] = c3.l[i]; }
    return r;
}
// (1 + u) a on component-split lanes: component 0 = a0 - a1, component 1 = a0 + a1
template <class X, int A, int N> HDF LV<2 * A, N> mul_xiB(const X &x, const LV<A, N> &a) {
    const auto ap = x.swap(a);
    return addB(a, x.pick_q(negB(ap), ap));
}

// Whole Fp values replicated on 4 lanes: lane r multiplies (a_r, b_r), every lane gets all four results.
template <class X, int A0, int A1, int A2, int A3, int B0, int B1, int B2, int B3, int N>
HDF Prod4<prod_bound((long)A0 * B0), prod_bound((long)A1 * B1), prod_bound((long)A2 * B2), prod_bound((long)A3 * B3), N>
round4_fp(const X &x, const LV<A0, N> &a0, const LV<A1, N> &a1, const LV<A2, N> &a2, const LV<A3, N> &a3,
          const LV<B0, N> &b0, const LV<B1, N> &b1, const LV<B2, N> &b2, const LV<B3, N> &b3) {
    static_assert(max4(A0, A1, A2, A3) <= kMaxK && max4(B0, B1, B2, B3) <= kMaxK, "product operand too large");
    const auto c = mulB(x.pick_r(a0, a1, a2, a3), x.pick_r(b0, b1, b2, b3));
    Prod4<prod_bound((long)A0 * B0), prod_bound((long)A1 * B1), prod_bound((long)A2 * B2), prod_bound((long)A3 * B3), N> r;
    const auto c0 = x.template from_lane<0>(c), c1 = x.template from_lane<1>(c), c2 = x.template from_lane<2>(c), c3 = x.template from_lane<3>(c);
    EIP_EACH_LANE { r.r0.l[i] = c0.l[i]; r.r1.l[i] = c1.l[i]; r.r2.l[i] = c2.l[i]; r.r3.l[i] = c3.l[i]; }
    return r;
}

}  // namespace eip
