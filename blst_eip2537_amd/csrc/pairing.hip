// Batched pairing check on gfx950.
//
// Replaces the reference's sequential loop (src/eip2537.c:1033-1068): per pair decode G1, G1
// subgroup test, decode G2, G2 subgroup test, Miller loop, running Fp12 product; then ONE final
// exponentiation for the whole batch (:1070) and the == 1 test (:1076).
//
// The product of Miller functions is reorganised so that no lane ever carries an Fp12 through
// the 63-step loop.  With l_{i,s} the line of pair i at step s (68 steps: 63 doublings, 5
// additions) the batch value is
//        F = prod_i f_i,   f_i = (...((l_{i,0})^2 l_{i,1})^2 ...)        =>
//        F = (...((L_0)^2 L_1)^2 ...),   L_s = prod_i l_{i,s}
// because squaring distributes over the product.  So:
//   k_pair_decode  [pair = 2 lanes]  wire decode + validation of P and Q, limb-form records, flags
//   k_pair_lines8  [pair = 8 lanes]  walk T = Q, 2Q, ... on the twist (homogeneous projective, two rounds of
//                           four Fp2 products per doubling), each lane holding one Fp2 component in limb form;
//                           stores the 68 sparse lines unscaled; the walk ends at T = [|z|]Q, which IS the
//                           G2 membership test psi(Q) == -[|z|]Q
//   k_pair_check_g1 [pair = 4 lanes]  G1 membership phi(P) == -[z^2]P  (second stream, beside the walk)
//   k_pair_fold    [a run of lines of one step per QUAD of lanes]  an Fp12 is spread over a quad (one Fp2
//                           component of three coefficients per lane); scaling of the records by (xP, yP),
//                           sparse line products, then pairwise dense products through shared memory
//   k_pair_tree2   [step]   the few per-block products of a step -> L_s in the host's layout
//   host                    63 squarings + 68 products over the L_s, conjugate, final
//                           exponentiation, == 1     (once per call, like the reference)
// Errors are merged with atomicMin on (pair << 4 | stage << 3 | code): lowest pair first, and
// inside a pair the reference's order G1 decode -> G1 subgroup -> G2 decode -> G2 subgroup.
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "codec.h"
#include "pairing.h"
#include "ifma.h"
#include "lanes.h"
#include "pairing_limb.h"
#include "dev_lanes.h"
#include "engine.h"

namespace eip {

#define HIPCHK(x)                                                                               \
    do {                                                                                        \
        hipError_t _e = (x);                                                                    \
        if (_e != hipSuccess) {                                                                 \
            fprintf(stderr, "[eip2537_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), \
                    __FILE__, __LINE__);                                                        \
            e->failed = true;          /* the slot is drained and rebuilt when the lease ends */        \
            return E_MEMORY_ERROR;                                                              \
        }                                                                                       \
    } while (0)

static constexpr int kPairWords = 96;    // 384 bytes
static constexpr int kSteps = 68;        // 63 doublings + 5 additions for |z| = 0xd201000000010000

// ---- records in HBM (limb form: 13 limbs of 30 bits, Montgomery factor R' = 2^390; 14 dwords apart) -------
struct PairPL { uint32_t x[14], y[14], xs[14], ys[14]; };      // P: x, y (membership), -3 x, 2 y (line scaling)
struct PairQL { uint32_t c[2][2][14]; };                       // Q: [component][x | y]
struct LineL { uint32_t v[2][3][14]; };                        // line of one pair at one step: [component][a0 | a1 | a4]
__device__ __forceinline__ void store_limbs(uint32_t *dst, const FpL &v) {
#pragma unroll
    for (int k = 0; k < 14; k += 2)
        *reinterpret_cast<uint2 *>(dst + k) = make_uint2(v.l[k], k + 1 < 13 ? v.l[k + 1] : 0u);
}
__device__ __forceinline__ FpL load_limbs(const uint32_t *src) {
    FpL v;
#pragma unroll
    for (int k = 0; k < 14; k += 2) {
        const uint2 t = *reinterpret_cast<const uint2 *>(src + k);
        v.l[k] = t.x;
        if (k + 1 < 13) v.l[k + 1] = t.y;
    }
    return v;
}
template <int K> __device__ __forceinline__ LV<K, 1> load_lv(const uint32_t *src) { return LV<K, 1>{{load_limbs(src)}}; }

// ---- lane groups on the device: one share per lane, exchanges are DPP moves (limbk.h has the host twins) ----
struct DevLanes4 {                       // 4 lanes holding whole Fp values, lane r computes product r  (G1 membership)
    int r;
    template <int J, int K> __device__ __forceinline__ LV<K, 1> from_lane(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xf, 0xf, true); })}};
    }
    template <int A, int B, int C, int D>
    __device__ __forceinline__ LV<max4(A, B, C, D), 1> pick_r(const LV<A, 1> &a, const LV<B, 1> &b, const LV<C, 1> &c, const LV<D, 1> &d) const {
        LV<max4(A, B, C, D), 1> o;
#pragma unroll
        for (int k = 0; k < 13; k++) o.l[0].l[k] = pick4(r, a.l[0].l[k], b.l[0].l[k], c.l[0].l[k], d.l[0].l[k]);
        return o;
    }
};
struct DevQuad {                         // 4 lanes (c, q) = index 2 c + q: half c of an Fp12, component q  (product tree)
    int c, q;
    template <int K> __device__ __forceinline__ LV<K, 1> swap(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return quad_perm<kDppSwap>(v); })}};
    }
    template <int Q, int K> __device__ __forceinline__ LV<K, 1> same_c(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return quad_perm<(Q == 0 ? kDppPair0 : kDppPair1)>(v); })}};
    }
    template <int Q, int K> __device__ __forceinline__ LV<K, 1> other_c(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return quad_perm<(Q == 0 ? 0x0A : 0x5F)>(v); })}};
    }
    template <int A, int B> __device__ __forceinline__ LV<max2(A, B), 1> pick_q(const LV<A, 1> &a, const LV<B, 1> &b) const {
        LV<max2(A, B), 1> r;
#pragma unroll
        for (int k = 0; k < 13; k++) r.l[0].l[k] = pick2(q == 0, a.l[0].l[k], b.l[0].l[k]);
        return r;
    }
    template <int A, int B> __device__ __forceinline__ LV<max2(A, B), 1> pick_c(const LV<A, 1> &a, const LV<B, 1> &b) const {
        LV<max2(A, B), 1> r;
#pragma unroll
        for (int k = 0; k < 13; k++) r.l[0].l[k] = pick2(c == 0, a.l[0].l[k], b.l[0].l[k]);
        return r;
    }
};

// Coalesced batches (several small calls' pairs back to back, api.hip): every call has its own
// first-error word; pair i belongs to the call j with coff[j] <= i < coff[j + 1].  A single call is the
// map with M = 1 (no table).
struct CallMap { const uint32_t *coff; int M; };
__device__ __forceinline__ void report_pair_error(unsigned long long *err, const CallMap &cm, uint32_t i, unsigned long long stage_code) {
    uint32_t j = 0, local = i;
    if (cm.M > 1) {
        int lo = 0, hi = cm.M;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cm.coff[mid] <= i) lo = mid; else hi = mid;
        }
        j = (uint32_t)lo;
        local = i - cm.coff[j];
    }
    atomicMin(&err[j], ((unsigned long long)local << 4) | stage_code);
}

// ---- wire decode, two lanes per pair ------------------------------------------------------------
// Lane 0 of a pair decodes and validates P (pad / < p / on curve), lane 1 does Q; both store the point in limb
// form (P also as the line factors -3 x, 2 y) and a "finite and valid" flag.  The walk and membership
// kernels then start from limbs and make no out-of-line calls at all.
// Error keys: (pair << 4 | stage << 3 | code), stage 0 = G1 (decode, then subgroup), 1 = G2 -- the
// reference's order inside a pair (src/eip2537.c:1036-1053); atomicMin keeps the first.
__global__ void __launch_bounds__(256)
k_pair_decode(const uint32_t *__restrict__ in, uint32_t k, PairPL *__restrict__ pl, PairQL *__restrict__ ql,
              uint8_t *__restrict__ flagP, uint8_t *__restrict__ flagQ, unsigned long long *err, CallMap cm) {
    auto dmul = [](const Fp &x, const Fp &y) { return fp_mul_cols(x, y); };                 // inlined (no out-of-line call in this kernel)
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 1;
    if (i >= k) return;
    if ((t & 1u) == 0) {
        Aff<Fp> P;
        const int st = decode_point_inl<Fp>(P, in + (size_t)i * kPairWords);
        if (st != E_SUCCESS) { report_pair_error(err, cm, i, (unsigned long long)st); P = Aff<Fp>{fp_zero(), fp_zero()}; }
        store_limbs(pl[i].x, to_limbs(dmul(P.x, Fp{{K_R390_MODP}})));
        store_limbs(pl[i].y, to_limbs(dmul(P.y, Fp{{K_R390_MODP}})));
        store_limbs(pl[i].xs, to_limbs(dmul(P.x, Fp{{K_R390_M3_MODP}})));
        store_limbs(pl[i].ys, to_limbs(dmul(P.y, Fp{{K_R390_2_MODP}})));
        flagP[i] = (st == E_SUCCESS && !is_inf(P)) ? 1 : 0;
    } else {
        Aff<Fp2> Q;
        const int st = decode_point_inl<Fp2>(Q, in + (size_t)i * kPairWords + 32);
        if (st != E_SUCCESS) { report_pair_error(err, cm, i, 8ull | (unsigned long long)st); Q = Aff<Fp2>{fp2_zero(), fp2_zero()}; }
        store_limbs(ql[i].c[0][0], to_limbs(dmul(Q.x.c0, Fp{{K_R390_MODP}})));
        store_limbs(ql[i].c[1][0], to_limbs(dmul(Q.x.c1, Fp{{K_R390_MODP}})));
        store_limbs(ql[i].c[0][1], to_limbs(dmul(Q.y.c0, Fp{{K_R390_MODP}})));
        store_limbs(ql[i].c[1][1], to_limbs(dmul(Q.y.c1, Fp{{K_R390_MODP}})));
        flagQ[i] = (st == E_SUCCESS && !is_inf(Q)) ? 1 : 0;
    }
}

// ---- G1 membership, 4 lanes per pair ----------------------------------------------------------
// phi(P) == -[z^2]P is two 64-bit double-and-add chains (126 doublings, 10 additions): the critical path of a
// batch beside the line walk.  A pair owns 4 lanes; a round of four Fp products goes one per lane (pairing_limb.h:
// homogeneous projective points, complete RCB formulas -- two rounds per doubling, three per addition, no
// special cases; values in limb form with compile-time bounds).
// EXCL: one wave per SIMD (batches whose walk + membership waves fit the chip's 1024 SIMDs)
template <bool EXCL>
__global__ void __launch_bounds__(64)
k_pair_check_g1(const PairPL *__restrict__ pl, const uint8_t *__restrict__ flagP, uint32_t k, unsigned long long *err, CallMap cm) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x >> 2);
    if (EXCL) claim_whole_simd();
    if (i >= k) return;                                       // uniform in the group
    if (!flagP[i]) return;                                    // infinity (a member) or already reported by the decode
    const DevLanes4 x{lane & 3};
    const LanePred<1> nm = g1_not_member_l(x, load_lv<1>(pl[i].x), load_lv<1>(pl[i].y));
    if (nm.b[0] && x.r == 0) report_pair_error(err, cm, i, (unsigned long long)E_NOT_IN_SUBGROUP);
}

// ---- line walk: 8 lanes per pair, split by Fp2 component ---------------------------------------
// A pair owns a group of 8 lanes = 4 lane pairs; lane (p, q) holds only COMPONENT q of every Fp2 value, the linear
// steps run on Fp, and a round multiplies four pairs of Fp2 operands -- lane pair p computes product p by the
// schoolbook rule, each lane one two-product sum with a single reduction (mul2L).  Round 3: T is homogeneous
// projective and a doubling step is TWO rounds (pairing_limb.h) where the Jacobian form took three, values stay
// in limb form (no re-slicing around the products, no conditional subtractions), and the square of X that the
// tangent needs is left to the product tree (throughput-bound, while this kernel is a latency chain).
// A record holds (a0, a1, a4) of pairing_limb.h's LineRecD / LineRecA; lane pair `part` stores coefficient `part`.
__device__ __forceinline__ void store_line_part(LineL *dst, int part, const FpL &v, const DevLanes8 &w) {
    if (w.p == part) store_limbs(dst->v[w.q][part], v);
}
__device__ __forceinline__ void store_identity_line(LineL *dst, const DevLanes8 &w) {
    if (w.p < 3) store_limbs(dst->v[w.q][w.p], (w.p == 0 && w.q == 0) ? fpl_one() : fpl_zero());
}
constexpr bool step_is_add(int s) { return s == 1 || s == 4 || s == 8 || s == 18 || s == 51; }       // after the doublings of bits 62, 60, 57, 48, 16

template <bool EXCL>
__global__ void __launch_bounds__(64)
k_pair_lines8(const PairQL *__restrict__ ql, const uint8_t *__restrict__ flagP, const uint8_t *__restrict__ flagQ,
              uint32_t k, LineL *__restrict__ lines, unsigned long long *err, CallMap cm) {
    const int lane = threadIdx.x & 63, sl = lane & 7;
    const uint32_t i = blockIdx.x * 8u + (uint32_t)(lane >> 3);
    if (EXCL) claim_whole_simd();              // batches whose walk + membership waves fit one per SIMD
    if (i >= k) return;                        // uniform within a lane group
    const DevLanes8 x{sl >> 1, sl & 1};
    const bool q_live = flagQ[i] != 0, contributes = q_live && flagP[i] != 0;          // else the pair contributes 1
    if (!q_live) {                             // Q at infinity (or undecodable: reported by the decode)
        for (int s = 0; s < kSteps; s++) store_identity_line(&lines[(size_t)s * k + i], x);
        return;
    }
    const LV<1, 1> qx = load_lv<1>(ql[i].c[x.q][0]), qy = load_lv<1>(ql[i].c[x.q][1]);
    WalkPt<1> T{widen<WalkK::X>(qx), widen<WalkK::Y>(qy), LV<WalkK::Z, 1>{{x.q ? fpl_zero() : fpl_one()}}};
    const uint64_t z = K_Z_ABS;
    int s = 0;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        {
            const LineRecD<1> l = miller_dbl_l(x, T);
            LineL *dst = &lines[(size_t)s * k + i];
            if (contributes) {
                store_line_part(dst, 0, l.a0.l[0], x);
                store_line_part(dst, 1, l.a1.l[0], x);
                store_line_part(dst, 2, l.a4.l[0], x);
            } else store_identity_line(dst, x);
            s++;
        }
        if ((z >> bit) & 1ull) {               // 5 of 63 steps
            const LineRecA<1> l = miller_add_l(x, T, qx, qy);
            LineL *dst = &lines[(size_t)s * k + i];
            if (contributes) {
                store_line_part(dst, 0, l.a0.l[0], x);
                store_line_part(dst, 1, l.a1.l[0], x);
                store_line_part(dst, 2, l.a4.l[0], x);
            } else store_identity_line(dst, x);
            s++;
        }
    }
    // T = [|z|]Q: the G2 membership test psi(Q) == -T
    const LanePred<1> nm = g2_not_member_l(x, T, qx, qy);
    if (nm.b[0] && sl == 0) report_pair_error(err, cm, i, 8ull | (unsigned long long)E_NOT_IN_SUBGROUP);
}

// ---- per-step line products --------------------------------------------------------------------
// L_s = prod_i l_{i,s}.  A QUAD of lanes holds an Fp12 (pairing_limb.h: lane (c, q) = component q of the even /
// odd coefficients) and folds its share of the step's lines into it one after the other: per line three Fp
// products to scale the stored record (X^2, times -3 xP; Y Z times 2 yP) and nine two-product sums, every lane
// busy (round 2 spread an Fp12 over 8 lanes of which 6 worked, on 12 x 32-bit words with 832 B of scratch per
// lane).  The quads' products are then multiplied pairwise through shared memory -- elements as 12 limb strings,
// a dense product dealt over 12 to 72 lanes (dense_terms) -- down to one element per block; k_pair_tree2 does
// the same over the blocks of a step and emits L_s in the host's 12 x 32-bit tower layout.
__device__ __forceinline__ int tower_slot(int k) { return (k & 1) * 3 + (k >> 1); }       // [c0.a0 c0.a1 c0.a2 c1.a0 c1.a1 c1.a2] = [w^0 w^2 w^4 w^1 w^3 w^5]

static constexpr int kTreeQuads = 64;                              // quads per block (256 threads)
static constexpr int kScratchLimbStrings = 256;                    // partial sums of a dense pass: (256 / LP) products x 12 outputs x PARTS <= 252
struct TreeShared {
    uint32_t elems[kTreeQuads * kElemWords];
    uint32_t scratch[kScratchLimbStrings * kLimbStride];
};
// one level of the pairwise product: element 2 i * stride *= element (2 i + 1) * stride, i < nprod
// the products of one level: product pr multiplies element left(pr) by element right(pr) in place (right == left squares it)
template <int UPL, class Left, class Right> __device__ __forceinline__ void dense_products(TreeShared &sh, int nprod, int tid, Left left, Right right) {
    constexpr int LP = 72 / UPL, PARTS = 6 / UPL, PER_PASS = 256 / LP;
    static_assert(UPL == 6 || PER_PASS * 12 * PARTS <= kScratchLimbStrings, "scratch too small");
    const int slot = tid / LP, idx = tid % LP, o = idx / PARTS, part = idx % PARTS;
    for (int p0 = 0; p0 < nprod; p0 += PER_PASS) {                 // uniform
        const int pr = p0 + slot;
        const bool active = slot < PER_PASS && pr < nprod;
        uint32_t *f = sh.elems + (size_t)left(active ? pr : 0) * kElemWords;
        const uint32_t *g = sh.elems + (size_t)right(active ? pr : 0) * kElemWords;
        FpL acc = fpl_zero();
        if (active) {
            acc = dense_terms<UPL>(f, g, o >> 1, o & 1, part * UPL).l[0];
            if (UPL < 6) {
                uint32_t *d = sh.scratch + (size_t)((slot * 12 + o) * PARTS + part) * kLimbStride;
#pragma unroll
                for (int t = 0; t < 13; t++) d[t] = acc.l[t];
            }
        }
        __syncthreads();                                           // operands read, partial sums visible
        if (active && part == 0) {
            if (UPL < 6) {
                for (int e = 1; e < PARTS; e++) {
                    const uint32_t *d = sh.scratch + (size_t)((slot * 12 + o) * PARTS + e) * kLimbStride;
                    FpL t;
#pragma unroll
                    for (int u = 0; u < 13; u++) t.l[u] = d[u];
                    acc = addL(acc, t);
                }
            }
            elem_store(f, o >> 1, o & 1, weak_reduceL(acc));       // <= 12 p  ->  <= 3 p
        }
        __syncthreads();
    }
}
template <class Left, class Right> __device__ __forceinline__ void dense_products_auto(TreeShared &sh, int nprod, int tid, Left left, Right right) {
    if (nprod > 10) dense_products<6>(sh, nprod, tid, left, right);          // lanes per product: as many as the block has for them
    else if (nprod > 7) dense_products<3>(sh, nprod, tid, left, right);
    else if (nprod > 3) dense_products<2>(sh, nprod, tid, left, right);
    else dense_products<1>(sh, nprod, tid, left, right);
}
// `regions` runs of `live` elements each, region r at element r * pitch: every run is multiplied down into its first element
__device__ __forceinline__ void block_tree(TreeShared &sh, int live, int tid, int regions = 1, int pitch = 0) {
    for (int stride = 1; stride < live; stride <<= 1) {
        const int ne = (live + stride - 1) / stride, per = ne / 2;
        dense_products_auto(sh, per * regions, tid, [=](int pr) { return (pr / per) * pitch + 2 * (pr % per) * stride; },
                            [=](int pr) { return (pr / per) * pitch + (2 * (pr % per) + 1) * stride; });
    }
}
// element -> the host's Fp12 (12 x 32-bit words, Montgomery factor 2^384, canonical), tower layout
__device__ __forceinline__ void emit_fp12(const uint32_t *elem, Fp2 *out, int tid) {
    if (tid < 12) {
        const int kq = tid >> 1, q = tid & 1;
        Fp *dst = reinterpret_cast<Fp *>(&out[tower_slot(kq)]) + q;
        *dst = fp_reduce_once(to_fpi(elem_load(elem, kq, q)).v);
    }
}
__device__ __forceinline__ void copy_elem(uint32_t *dst, const uint32_t *src, int tid) {
    for (int t = tid; t < kElemWords; t += 256) dst[t] = src[t];
}

// grid (blocks, 68 steps) -- or (calls, 68) for a coalesced batch --, 256 threads = 64 quads.  Quad G of a step folds
// the lines G, G + NQ, G + 2 NQ, ... (NQ quads per step) of the range [base, base + count).
// FINAL: the block's product IS L_s (one block per step / per (call, step)): emitted in the host layout to
// out_fp12[blockIdx.x * 68 + step]; otherwise the block's element goes to blk_out[step][block].
#ifndef EIP_FOLD_WAVES
#define EIP_FOLD_WAVES 2
#endif
// FUSED (round 4): the six-product form of the line product (quad_fold_line<true>), one wave per SIMD (349 registers, no scratch, 3 blocks per
// step): 0.53 -> 0.50 ms for the fold + tree of a 2^12-pair check; larger checks are throughput-bound and keep two waves per SIMD
// (2^16 pairs: 4.71 ms against 4.79 fused; the fused form AT two waves per SIMD spills 93 registers and loses everywhere)
template <bool BATCH, bool FUSED>
__global__ void __launch_bounds__(256, FUSED ? 1 : EIP_FOLD_WAVES)
k_pair_fold(const LineL *__restrict__ lines, const PairPL *__restrict__ pl, uint32_t k, const uint32_t *__restrict__ coff,
            uint32_t *__restrict__ blk_out, Fp2 *__restrict__ out_fp12, int final_out) {
    __shared__ TreeShared sh;
    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, qd = tid >> 2;
    const DevQuad x{(lane >> 1) & 1, lane & 1};
    uint32_t base = 0, count = k, nq = gridDim.x * kTreeQuads, G = blockIdx.x * kTreeQuads + qd;
    if (BATCH) { base = coff[blockIdx.x]; count = coff[blockIdx.x + 1] - base; nq = kTreeQuads; G = qd; }
    const bool is_add = step_is_add(s);
    Fp12Q<1> f;
    f.own[0] = LV<TreeK::F, 1>{{(x.c == 0 && x.q == 0) ? fpl_one() : fpl_zero()}};
    f.own[1] = LV<TreeK::F, 1>{{fpl_zero()}};
    f.own[2] = LV<TreeK::F, 1>{{fpl_zero()}};
    bool have = false;
    for (uint32_t li = G; li < count; li += nq) {               // uniform within the quad
        const uint32_t i = base + li;
        const LineL *rec = &lines[(size_t)s * k + i];
        // the three coefficients are loaded and scaled one after the other (the empty asm keeps the loads from being hoisted
        // together: five limb strings in flight at once cost the loop more spills at two waves per SIMD)
        const auto r1 = load_lv<LineK::A1A>(rec->v[x.q][1]);
        const auto xs = load_lv<1>(pl[i].xs);
        // a1 = X^2 xs (tangent) or (2 theta) xs (chord); component 0 of X^2 = (x0 + x1)(x0 - x1), component 1 = 2 x0 x1
        const auto rp = x.swap(r1);
        const auto sq = mulB(x.pick_q(addB(r1, rp), dblB(r1)), x.pick_q(subB(r1, rp), rp));
        LV<max2(decltype(sq)::kK, LineK::A1A), 1> pre;
#pragma unroll
        for (int t = 0; t < 13; t++) pre.l[0].l[t] = pick2(is_add, r1.l[0].l[t], sq.l[0].l[t]);      // by value: an lvalue conditional would pin both in scratch
        const auto a1 = mulB(pre, xs);
        asm volatile("" ::: "memory");
        const auto r4 = load_lv<LineK::A4A>(rec->v[x.q][2]);
        const auto ys = load_lv<1>(pl[i].ys);
        const auto a4 = mulB(r4, ys);
        asm volatile("" ::: "memory");
        const auto a0 = load_lv<LineK::A0>(rec->v[x.q][0]);
        if (!have) { f = quad_seed_line(x, a0, a1, a4); have = true; }
        else quad_fold_line<FUSED>(x, f, a0, a1, a4);
    }
    // quad -> element in shared memory (natural [k][q] order; the odd half is held rotated by one)
    {
        uint32_t *e = sh.elems + (size_t)qd * kElemWords;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int kk = x.c == 0 ? 2 * t : 2 * ((t + 1) % 3) + 1;
            elem_store(e, kk, x.q, weak_reduceL(f.own[t].l[0]));
        }
    }
    __syncthreads();
    const uint32_t first = BATCH ? 0u : blockIdx.x * kTreeQuads;
    const int live = (int)min((uint32_t)kTreeQuads, count > first ? count - first : 0u);
    block_tree(sh, max(live, 1), tid);
    if (final_out) emit_fp12(sh.elems, out_fp12 + ((size_t)blockIdx.x * kSteps + s) * 6, tid);
    else copy_elem(blk_out + ((size_t)s * gridDim.x + blockIdx.x) * kElemWords, sh.elems, tid);
}

// One block per GROUP of kGroupSteps consecutive Miller steps: the per-block elements of each step are multiplied down to
// L_s (the steps of the group side by side), and then the group's own share of the host's Horner pass is done here:
//     M = prod_i L_{s_i}^(2^{e_i}),   e_i = doublings after step s_i inside the group       (acc = acc^2 . L for a doubling step)
// so that the host squares 63 times as before but multiplies 17 times instead of 68 (F = F^(2^E) M per group): the
// products it no longer does were a third of its 0.6 ms; here they are a few dense products on 72 lanes each.
static constexpr int kGroupSteps = 4, kGroups = (kSteps + kGroupSteps - 1) / kGroupSteps, kGroupPitch = 8;
__global__ void __launch_bounds__(256)
k_pair_tree2(const uint32_t *__restrict__ blk_out, uint32_t nblk, Fp2 *__restrict__ group_out) {
    __shared__ TreeShared sh;
    const int g = blockIdx.x, tid = threadIdx.x, s0 = g * kGroupSteps, m = min(kGroupSteps, kSteps - s0);
    for (int r = 0; r < m; r++)
        for (uint32_t t = tid; t < nblk * kElemWords; t += 256)
            sh.elems[(size_t)r * kGroupPitch * kElemWords + t] = blk_out[(size_t)(s0 + r) * nblk * kElemWords + t];
    __syncthreads();
    block_tree(sh, (int)nblk, tid, m, kGroupPitch);                         // element r * pitch = L_{s0 + r}
    for (int r = 1; r < m; r++) {                                           // uniform
        if (!step_is_add(s0 + r)) dense_products<1>(sh, 1, tid, [](int) { return 0; }, [](int) { return 0; });
        dense_products<1>(sh, 1, tid, [](int) { return 0; }, [=](int) { return r * kGroupPitch; });
    }
    emit_fp12(sh.elems, group_out + (size_t)g * 6, tid);
}

// Decode, membership and line walk of K pairs that belong to M calls (M = 1: one call, no table): the
// part of the device pipeline a single call and a coalesced batch share.  `d_coff` is the device copy of
// the call offsets (nullptr for M = 1).  Leaves ev_a / ev_b around the walk and the fork joined.
struct PairBufs { unsigned long long *err; LineL *lines; PairPL *pl; PairQL *ql; uint8_t *flagP, *flagQ; };
static int pairing_reserve(Engine *e, size_t k, size_t err_bytes, size_t winout_bytes, PairBufs &b) {
    HIPCHK(e->misc.reserve(err_bytes));
    HIPCHK(e->pts.reserve(k * sizeof(PairPL)));
    HIPCHK(e->digits.reserve(k * sizeof(PairQL) + 2 * k + 64));        // decoded Q of every pair, then the two flag arrays
    HIPCHK(e->partial.reserve((size_t)kSteps * k * sizeof(LineL)));
    HIPCHK(e->winout.reserve(winout_bytes));
    b.lines = reinterpret_cast<LineL *>(e->partial.p);
    b.pl = reinterpret_cast<PairPL *>(e->pts.p);
    b.ql = reinterpret_cast<PairQL *>(e->digits.p);
    b.flagP = reinterpret_cast<uint8_t *>(b.ql + k);
    b.flagQ = b.flagP + k;
    return E_SUCCESS;
}
static int pairing_front(Engine *e, const uint32_t *in, size_t k, const uint32_t *d_coff, int M, const PairBufs &b) {
    const uint32_t line_blocks = (uint32_t)((k + 7) / 8), check_blocks = (uint32_t)((k + 15) / 16);
    const bool excl = line_blocks + check_blocks <= chip_shape(e->device).simds;       // one wave per SIMD while everything fits the chip (1 024 SIMDs on a whole MI355X)
    {
        LastPlan lp{};
        snprintf(lp.kernel, sizeof lp.kernel, "%s", excl ? "k_pair_lines8<true>" : "k_pair_lines8<false>");
        lp.windows = kSteps; lp.lanes = 8; lp.units = (uint32_t)k; lp.shards = 1;
        e->last_plan = lp;
    }
    const CallMap cm{d_coff, M};
    hipStream_t s = e->stream;
    HIPCHK(hipMemsetAsync(b.err, 0xFF, (size_t)M * 8, s));
    HIPCHK(hipEventRecord(e->ev_start, s));
    hipLaunchKernelGGL(k_pair_decode, dim3((uint32_t)((2 * k + 255) / 256)), dim3(256), 0, s, in, (uint32_t)k, b.pl, b.ql, b.flagP, b.flagQ, b.err, cm);
    HIPCHK(hipEventRecord(e->ev_j3, s));
    // fork: the G1 membership kernel runs beside the line walk
    HIPCHK(e->need_stream2());
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_j3, 0));
    if (excl)
        hipLaunchKernelGGL(k_pair_check_g1<true>, dim3(check_blocks), dim3(64), 0, e->stream2, b.pl, b.flagP, (uint32_t)k, b.err, cm);
    else
        hipLaunchKernelGGL(k_pair_check_g1<false>, dim3(check_blocks), dim3(64), 0, e->stream2, b.pl, b.flagP, (uint32_t)k, b.err, cm);
    HIPCHK(hipEventRecord(e->ev_j2, e->stream2));
    HIPCHK(hipEventRecord(e->ev_a, s));
    if (excl) hipLaunchKernelGGL(k_pair_lines8<true>, dim3(line_blocks), dim3(64), 0, s, b.ql, b.flagP, b.flagQ, (uint32_t)k, b.lines, b.err, cm);
    else hipLaunchKernelGGL(k_pair_lines8<false>, dim3(line_blocks), dim3(64), 0, s, b.ql, b.flagP, b.flagQ, (uint32_t)k, b.lines, b.err, cm);
    HIPCHK(hipEventRecord(e->ev_b, s));
    return E_SUCCESS;
}

int pairing_device(Engine *e, const void *d_in, size_t k, uint32_t *ml_words) {
    if (k == 0 || k >= (1ull << 27)) return E_MEMORY_ERROR;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) != 0) {
        fprintf(stderr, "[eip2537_hip] device input must be 4-byte aligned\n");
        return E_MEMORY_ERROR;
    }
    // blocks per step: as many as keep the whole grid (blocks x 68 steps) in ONE round of 2 blocks per CU (512
    // slots); one block more than that and the kernel takes two block-times
    //  k_pair_tree2 stacks the per-block elements of a step at a pitch of kGroupPitch inside kTreeQuads slots
    static_assert((256 * EIP_FOLD_WAVES) / kSteps <= kGroupPitch, "k_pair_tree2: blocks per step exceed the group pitch");
    static_assert(kGroupSteps * kGroupPitch <= kTreeQuads, "k_pair_tree2: a group's elements exceed the block's slots");
    const bool fused = k <= 8192u;               // the six-product fold at one wave per SIMD (profiles/r04_pairing_fold.txt)
    const uint32_t max_blocks_per_step = std::max(1u, std::min<uint32_t>((uint32_t)kGroupPitch, (chip_shape(e->device).cus * (fused ? 1u : (uint32_t)EIP_FOLD_WAVES)) / kSteps));     // 3 | 7 on 256 CUs
    const uint32_t tree_blocks = (uint32_t)std::min<size_t>(max_blocks_per_step, (k + kTreeQuads - 1) / kTreeQuads);
    PairBufs b{};
    int st = pairing_reserve(e, k, 64, (size_t)kSteps * tree_blocks * kElemWords * 4 + (size_t)kSteps * sizeof(Fp12), b);
    if (st) return st;
    b.err = reinterpret_cast<unsigned long long *>(e->misc.p);
    auto *blk_out = reinterpret_cast<uint32_t *>(e->winout.p);                            // [step][block] limb-form elements
    auto *step_out = reinterpret_cast<Fp2 *>(blk_out + (size_t)kSteps * tree_blocks * kElemWords);      // [step] Fp12, host layout
    hipStream_t s = e->stream;
    st = pairing_front(e, reinterpret_cast<const uint32_t *>(d_in), k, nullptr, 1, b);
    if (st) return st;
    if (fused)
        hipLaunchKernelGGL((k_pair_fold<false, true>), dim3(tree_blocks, kSteps), dim3(256), 0, s, b.lines, b.pl, (uint32_t)k, (const uint32_t *)nullptr,
                           blk_out, step_out, 0);
    else
    hipLaunchKernelGGL((k_pair_fold<false, false>), dim3(tree_blocks, kSteps), dim3(256), 0, s, b.lines, b.pl, (uint32_t)k, (const uint32_t *)nullptr,
                       blk_out, step_out, 0);
    hipLaunchKernelGGL(k_pair_tree2, dim3(kGroups), dim3(256), 0, s, blk_out, tree_blocks, step_out);     // step_out: one Fp12 per group of steps
    HIPCHK(hipEventRecord(e->ev_c, s));
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());

    // first-error word | the 17 group products, into the slot's pinned buffer (round 4: no staging copy on the way back)
    HIPCHK(e->need_pinned(64 + (size_t)kGroups * sizeof(Fp12)));
    unsigned long long *herr_p = static_cast<unsigned long long *>(e->pinned);
    Fp12 *Lp = reinterpret_cast<Fp12 *>(static_cast<char *>(e->pinned) + 64);
    StreamDrain drain{s};
    HIPCHK(hipMemcpyAsync(herr_p, b.err, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(Lp, step_out, (size_t)kGroups * sizeof(Fp12), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    drain.armed = false;
    const unsigned long long herr = *herr_p;
    std::vector<Fp12> L(Lp, Lp + kGroups);                    // (aligned copy for the vector code below: 10 KB)
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_j3, e->ev_j2) == hipSuccess) e->last_aux_ms[0] = ms;      // decode done -> membership done
    if (hipEventElapsedTime(&ms, e->ev_b, e->ev_c) == hipSuccess) e->last_aux_ms[1] = ms;        // walk done -> L_s written
    if (herr != ~0ull) return (int)(herr & 7ull);
    // F = F^(2^E) M per group, E = the doubling steps of the group (63 squarings and 17 products in all)
    Fp12 F = fp12_one();
    int nsq[kGroups];
    for (int g = 0; g < kGroups; g++) {
        nsq[g] = 0;
        for (int s2 = g * kGroupSteps; s2 < std::min(kSteps, (g + 1) * kGroupSteps); s2++) nsq[g] += step_is_add(s2) ? 0 : 1;
    }
#if defined(EIP_HAVE_IFMA)
    if (host_ifma_enabled()) F = ifma::horner_groups_ifma(L.data(), nsq, kGroups);          // AVX-512 IFMA: 0.4 us per Fp12 product
    else
#endif
    for (int g = 0; g < kGroups; g++) {
        for (int k2 = 0; k2 < nsq[g] && g > 0; k2++) F = sqr(F);          // (the squarings before the first group are of one)
        F = g == 0 ? L[0] : mul(F, L[(size_t)g]);
    }
    F = conj(F);
    memcpy(ml_words, &F, sizeof F);
    return E_SUCCESS;
}

// Coalesced batch of M small pairing calls (each at most kPairBatchMaxPairs pairs; api.hip): one decode /
// membership / walk over the concatenated pairs with a first-error word per call, one product-tree block
// per (call, step).  Writes rc[j] and, for the good calls, their 68 per-step products to L (the caller
// finishes with miller_product_from_steps and the final exponentiation on its own thread).
int pairing_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *L_words, int *rc) {
    const size_t k = coff[M];
    if (M < 1 || M > kPairBatchMaxCalls || k == 0 || k > 8192) return E_MEMORY_ERROR;
    uint32_t kmax = 0;
    for (int j = 0; j < M; j++) kmax = std::max(kmax, coff[j + 1] - coff[j]);
    if (kmax == 0 || kmax > (uint32_t)kPairBatchMaxPairs) return E_MEMORY_ERROR;
    PairBufs b{};
    int st = pairing_reserve(e, k, 64 + (size_t)M * 8 + (size_t)(M + 1) * 4, (size_t)M * kSteps * sizeof(Fp12), b);
    if (st) return st;
    b.err = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(e->misc.p) + 64);                  // [M]
    auto *d_coff = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 64 + (size_t)M * 8);     // [M + 1]
    auto *step_out = reinterpret_cast<Fp2 *>(e->winout.p);                // [call][step] Fp12
    hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(d_coff, coff, (size_t)(M + 1) * 4, hipMemcpyHostToDevice, s));
    st = pairing_front(e, reinterpret_cast<const uint32_t *>(d_in), k, d_coff, M, b);
    if (st) return st;
    hipLaunchKernelGGL((k_pair_fold<true, false>), dim3((uint32_t)M, kSteps), dim3(256), 0, s, b.lines, b.pl, (uint32_t)k, (const uint32_t *)d_coff,
                       (uint32_t *)nullptr, step_out, 1);
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> herr((size_t)M);
    StreamDrain drain{s};
    HIPCHK(hipMemcpyAsync(herr.data(), b.err, (size_t)M * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L_words, step_out, (size_t)M * kSteps * sizeof(Fp12), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    drain.armed = false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    for (int j = 0; j < M; j++) rc[j] = herr[(size_t)j] != ~0ull ? (int)(herr[(size_t)j] & 7ull) : E_SUCCESS;
    return E_SUCCESS;
}

}  // namespace eip
