// Wavefront (64-lane) data movement for the lane-group kernels: field elements and points moved
// between lanes with ds_bpermute shuffles, and per-lane operand selection.  Used where one logical
// operation is spread over 2, 4 or 8 lanes (k_msm_accum2, k_msm_reduce4, k_pair_lines, k_pair_tree).
#pragma once
#include "curve.h"

namespace eip {

// One wave per SIMD.  The chain-bound kernels (pairing line walk and G1 membership, MSM bucket reduce)
// run few waves, each bound by its own instruction stream; two of them on one SIMD share its issue
// slots and both run at ~0.7 of their speed while other SIMDs sit idle -- and the dispatcher does place
// them so whenever their register counts allow (measured at 2^12 pairs once the walk stopped needing
// AGPRs and could co-reside with a membership wave: per-instruction time +25 % / +45 %,
// k_pair_check_g1 1.12 -> 1.60 ms; with the claim 1.07 ms and the walk 1.73 -> 1.46 ms).
// Touching the last accumulation register makes a kernel's allocation exceed half of the SIMD's 512
// registers, so the hardware cannot place a second such wave there and spreads the blocks instead.
__device__ __forceinline__ void claim_whole_simd() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_accvgpr_write_b32 a127, 0" ::: "a127");
#endif
}

__device__ __forceinline__ Fp shfl_from(const Fp &a, int src) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = __shfl(a.l[i], src, 64);
    return r;
}
__device__ __forceinline__ FpI shfl_from(const FpI &a, int src) { return FpI{shfl_from(a.v, src)}; }
__device__ __forceinline__ Fp2 shfl_from(const Fp2 &a, int src) { return Fp2{shfl_from(a.c0, src), shfl_from(a.c1, src)}; }
template <class T> __device__ __forceinline__ Xyzz<T> shfl_from(const Xyzz<T> &p, int src) {
    return Xyzz<T>{shfl_from(p.x, src), shfl_from(p.y, src), shfl_from(p.zz, src), shfl_from(p.zzz, src)};
}
// Value of lane J of the caller's aligned 4-lane group: a DPP quad_perm broadcast -- one VALU move per dword,
// no LDS round trip (ds_bpermute) in the chain-bound 4-lane kernels.  EIP_QUAD_DPP=0 builds the shuffle form.
#ifndef EIP_QUAD_DPP
#define EIP_QUAD_DPP 1
#endif
template <int J> __device__ __forceinline__ Fp quad_from(const Fp &a) {
    Fp r;
#if EIP_QUAD_DPP
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], J * 0x55, 0xf, 0xf, true);
#else
    const int src = (threadIdx.x & 60) + J;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = __shfl(a.l[i], src, 64);
#endif
    return r;
}
template <int J> __device__ __forceinline__ FpI quad_from(const FpI &a) { return FpI{quad_from<J>(a.v)}; }
// the same for an arbitrary quad permutation: PAIR0 / PAIR1 = lane 0 / 1 of the caller's aligned lane pair,
// SWAP = the partner lane (lane ^ 1)
static constexpr int kDppPair0 = 0xA0, kDppPair1 = 0xF5, kDppSwap = 0xB1;
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
#if EIP_QUAD_DPP
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
#else
    const int l = threadIdx.x & 63;
    return __shfl(v, (l & 60) + ((CTRL >> (2 * (l & 3))) & 3), 64);
#endif
}
template <int CTRL> __device__ __forceinline__ Fp quad_perm(const Fp &a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = quad_perm<CTRL>(a.l[i]);
    return r;
}
template <int CTRL> __device__ __forceinline__ FpI quad_perm(const FpI &a) { return FpI{quad_perm<CTRL>(a.v)}; }
template <int CTRL> __device__ __forceinline__ Fp2 quad_perm(const Fp2 &a) { return Fp2{quad_perm<CTRL>(a.c0), quad_perm<CTRL>(a.c1)}; }
// Value of lane 2 J + (own parity) of the caller's aligned 8-lane group (J = 0 .. 3: the lane pair): the pair is
// first broadcast inside each quad, then the quad that does not hold it takes the other quad's copy with a
// row shift by 4 restricted to its own banks -- two VALU moves per dword instead of a ds_bpermute.
template <int J> __device__ __forceinline__ uint32_t group8_pair(uint32_t v) {
#if EIP_QUAD_DPP
    const int t = __builtin_amdgcn_mov_dpp((int)v, (J & 1) ? 0xEE : 0x44, 0xf, 0xf, true);
    if (J < 2) return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xA, false);    // row_shr:4 into lanes 4..7, 12..15
    return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);               // row_shl:4 into lanes 0..3, 8..11
#else
    const int l = threadIdx.x & 63;
    return __shfl(v, (l & 56) + 2 * J + (l & 1), 64);
#endif
}
template <int J> __device__ __forceinline__ FpI group8_pair(const FpI &a) {
    FpI r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.v.l[i] = group8_pair<J>(a.v.l[i]);
    return r;
}

template <int J> __device__ __forceinline__ Fp2 quad_from(const Fp2 &a) { return Fp2{quad_from<J>(a.c0), quad_from<J>(a.c1)}; }
// value of lane + off (lanes past the end read their own value; callers mask them out)
template <class T> __device__ __forceinline__ T shfl_down(const T &a, int off) {
    const int lane = threadIdx.x & 63;
    return shfl_from(a, lane + off < 64 ? lane + off : lane);
}

// Selections go through by-value helpers on purpose: "r == 0 ? a.l[i] : b.l[i]" on lvalues is an lvalue
// conditional -- the compiler selects the ADDRESS and loads afterwards, which pins a, b, ... in scratch
// memory (k_msm_reduce4 carried 1.1 KB of scratch per lane and wrote 580 MB per launch that way).
__device__ __forceinline__ uint32_t pick2(bool first, uint32_t a, uint32_t b) { return first ? a : b; }
__device__ __forceinline__ uint32_t pick4(int r, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return r == 0 ? a : r == 1 ? b : r == 2 ? c : d;
}
__device__ __forceinline__ Fp sel4(int r, const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
    Fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = pick4(r, a.l[i], b.l[i], c.l[i], d.l[i]);
    return o;
}
__device__ __forceinline__ FpI sel4(int r, const FpI &a, const FpI &b, const FpI &c, const FpI &d) { return FpI{sel4(r, a.v, b.v, c.v, d.v)}; }
__device__ __forceinline__ Fp2 sel4(int r, const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) {
    return Fp2{sel4(r, a.c0, b.c0, c.c0, d.c0), sel4(r, a.c1, b.c1, c.c1, d.c1)};
}
__device__ __forceinline__ Fp sel2(int r, const Fp &a, const Fp &b) {
    Fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = pick2(r == 0, a.l[i], b.l[i]);
    return o;
}
__device__ __forceinline__ FpI sel2(int r, const FpI &a, const FpI &b) { return FpI{sel2(r, a.v, b.v)}; }
__device__ __forceinline__ Fp2 sel2(int r, const Fp2 &a, const Fp2 &b) {
    Fp2 o;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        o.c0.l[i] = pick2(r == 0, a.c0.l[i], b.c0.l[i]);
        o.c1.l[i] = pick2(r == 0, a.c1.l[i], b.c1.l[i]);
    }
    return o;
}

// ---- Fp2 values split by component over lane pairs ---------------------------------------------------
// The lane-group kernels over Fp2 (pairing line walk, G2 bucket reduce) give one logical operation a group
// of 8 lanes = 4 lane pairs.  Lane (p, q) holds only COMPONENT q of every Fp2 value of the group
// (replicated over the four pairs p): linear steps are component-wise, so they run on Fp -- half the
// instructions and registers of lanes that hold whole Fp2 values -- and only a product needs the partner
// lane's component of its two operands (one exchange with lane ^ 1).  A round multiplies four pairs of
// operands: lane pair p computes product p by the schoolbook rule (c0 = a0 b0 - a1 b1,
// c1 = a0 b1 + a1 b0: two Fp products per lane), and every lane gets its component of all four results.
// Components are FpI values: kept in [0, 2p) (field.h), so a product needs no final conditional
// subtraction; whatever leaves a kernel goes through fp_canon().
struct Prod4c { FpI r0, r1, r2, r3; };            // this lane's component of the four products of a round
struct PairProd8 {
    int p, q, lane, gbase;
    __device__ __forceinline__ PairProd8(int lane_, int sl, int gb) : p(sl >> 1), q(sl & 1), lane(lane_), gbase(gb) {}
    __device__ __forceinline__ Prod4c operator()(const FpI &a0, const FpI &a1, const FpI &a2, const FpI &a3,
                                                 const FpI &b0, const FpI &b1, const FpI &b2, const FpI &b3) const {
        const FpI u = sel4(p, a0, a1, a2, a3), v = sel4(p, b0, b1, b2, b3);     // own components of this pair's operands
        const FpI up = quad_perm<kDppSwap>(u), vp = quad_perm<kDppSwap>(v);         // the partner's
        // q = 0: c0 = u0 v0 - u1 v1      q = 1: c1 = u0 v1 + u1 v0
        // one two-product sum with a single reduction (field.h, fp_mul2_cols30); the difference as u1 * (2p - v1)
        const FpI c = mul2(sel2(q, u, up), v, sel2(q, up, u), sel2(q, neg(vp), vp));
        return Prod4c{group8_pair<0>(c), group8_pair<1>(c), group8_pair<2>(c), group8_pair<3>(c)};
    }
    // an Fp2 predicate holds when it holds on both components
    __device__ __forceinline__ bool both(bool mine) const { const uint32_t m = mine ? 1u : 0u; return (m & quad_perm<kDppSwap>(m)) != 0; }
};
// A point over Fp2 seen from one lane: Xyzz<FpI> holding this lane's component of each coordinate.
__device__ __forceinline__ Xyzz<FpI> component_of(const Xyzz<Fp2> &p, int q) {
    return Xyzz<FpI>{FpI{sel2(q, p.x.c0, p.x.c1)}, FpI{sel2(q, p.y.c0, p.y.c1)}, FpI{sel2(q, p.zz.c0, p.zz.c1)}, FpI{sel2(q, p.zzz.c0, p.zzz.c1)}};
}
// lane q of a pair writes component q of every coordinate, canonical
__device__ __forceinline__ void store_component(Xyzz<Fp2> *dst, const Xyzz<FpI> &p, int q) {
    Fp *out = reinterpret_cast<Fp *>(dst);
    out[0 + q] = fp_canon(p.x); out[2 + q] = fp_canon(p.y); out[4 + q] = fp_canon(p.zz); out[6 + q] = fp_canon(p.zzz);
}
__device__ __forceinline__ bool is_inf8c(const Xyzz<FpI> &p, const PairProd8 &pp) { return pp.both(is_zero(p.zz)); }
// 2P (dbl-2008-s-1), three rounds; infinity stays infinity (zz = 0 propagates)
__device__ __forceinline__ Xyzz<FpI> dbl8c(const Xyzz<FpI> &p, const PairProd8 &prod) {
    const FpI U = dbl(p.y);
    Prod4c pr = prod(U, p.x, U, U, U, p.x, U, U);
    const FpI V = pr.r0, XX = pr.r1;
    const FpI M = add(dbl(XX), XX);
    pr = prod(U, p.x, M, V, V, V, M, p.zz);
    const FpI W = pr.r0, S = pr.r1, MM = pr.r2, ZZ3 = pr.r3;
    const FpI X3 = sub(MM, dbl(S));
    pr = prod(M, W, W, W, sub(S, X3), p.y, p.zzz, p.zzz);
    return Xyzz<FpI>{X3, sub(pr.r0, pr.r1), ZZ3, pr.r2};
}
// P + Q (add-2008-s), four rounds, complete
__device__ __forceinline__ Xyzz<FpI> add8c(const Xyzz<FpI> &p, const Xyzz<FpI> &q, const PairProd8 &prod) {
    if (is_inf8c(q, prod)) return p;                           // uniform in the group
    if (is_inf8c(p, prod)) return q;
    Prod4c pr = prod(p.x, q.x, p.y, q.y, q.zz, p.zz, q.zzz, p.zzz);
    const FpI U1 = pr.r0, U2 = pr.r1, S1 = pr.r2, S2 = pr.r3;
    const FpI Pd = sub(U2, U1), Rr = sub(S2, S1);
    if (prod.both(is_zero(Pd))) {                              // same x: double or cancel (rare)
        if (prod.both(is_zero(Rr))) return dbl8c(p, prod);
        return xyzz_inf<FpI>();
    }
    pr = prod(Pd, Rr, p.zz, p.zzz, Pd, Rr, q.zz, q.zzz);
    const FpI PP = pr.r0, RR = pr.r1, ZZ12 = pr.r2, ZZZ12 = pr.r3;
    pr = prod(Pd, U1, ZZ12, ZZ12, PP, PP, PP, PP);
    const FpI PPP = pr.r0, Qv = pr.r1, ZZ3 = pr.r2;
    const FpI X3 = sub(sub(RR, PPP), dbl(Qv));
    pr = prod(Rr, S1, ZZZ12, ZZZ12, sub(Qv, X3), PPP, PPP, PPP);
    return Xyzz<FpI>{X3, sub(pr.r0, pr.r1), ZZ3, pr.r2};
}
__device__ __forceinline__ Xyzz<FpI> small_mul8c(const Xyzz<FpI> &p, uint32_t m, const PairProd8 &prod) {
    if (m == 0) return xyzz_inf<FpI>();
    Xyzz<FpI> acc = p;                               // the top bit: no doubling of the point at infinity
    for (int i = 30 - __builtin_clz(m); i >= 0; i--) {
        acc = dbl8c(acc, prod);
        if ((m >> i) & 1u) acc = add8c(acc, p, prod);
    }
    return acc;
}

}  // namespace eip
