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
//   k_pair_decode  [pair = 2 lanes]  wire decode + validation of P and Q, Montgomery form, flags
//   k_pair_lines8  [pair = 8 lanes]  walk T = Q, 2Q, ... on the twist, the independent products of a
//                           step dealt over the lane pairs, each lane holding one Fp2 component; stores
//                           the 68 sparse lines (a0, a1, a4 -- scaled by xP, yP later); the walk ends at
//                           T = [|z|]Q, which IS the G2 membership test psi(Q) == -[|z|]Q
//   k_pair_check_g1 [pair = 4 lanes]  G1 membership phi(P) == -[z^2]P  (second stream, beside the walk)
//   k_pair_tree    [a run of lines of one step per 8-lane group]  each Fp12 is spread over a lane
//                           group (one Fp2 coefficient per lane); sparse line products, then a
//                           per-wave product tree over shuffles and an LDS step across waves
//   k_pair_tree2   [step]   the few per-block products of a step -> L_s (same lane-group form)
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
#include "lanes.h"
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

struct LineRec { Fp2 a0, a1, a4; };      // l = a0 + a1 v + a4 v w   (a1, a4 already scaled by xP, yP)

// ---- G1 membership, 4 lanes per pair ----------------------------------------------------------
// phi(P) == -[z^2]P is two 64-bit double-and-add chains (126 doublings, 10 additions).  One lane per
// pair made that 1.5-1.8 ms of serial products -- the critical path of every batch below ~2000 pairs
// once the line walk got shorter.  Here a pair owns 4 lanes: the independent Fp products of an
// XYZZ doubling (3 rounds) or addition (4 rounds) go one per lane and are exchanged by shuffles.
// (values are FpI: kept in [0, 2p), no conditional subtraction after a product -- field.h)
// P + Q (add-2008-s), operands replicated on the 4 lanes of the group, complete
__device__ __forceinline__ Xyzz<FpI> g1_add4(const Xyzz<FpI> &p, const Xyzz<FpI> &q, int r, int gb) {
    if (is_inf(q)) return p;                                  // uniform in the group
    if (is_inf(p)) return q;
    FpI pr = mul(sel4(r, p.x, q.x, p.y, q.y), sel4(r, q.zz, p.zz, q.zzz, p.zzz));
    const FpI U1 = quad_from<0>(pr), U2 = quad_from<1>(pr), S1 = quad_from<2>(pr), S2 = quad_from<3>(pr);
    const FpI Pd = sub(U2, U1), Rr = sub(S2, S1);
    if (is_zero(Pd)) {                                        // same x: double or cancel (small-order inputs)
        if (is_zero(Rr)) return dbl(p);
        return xyzz_inf<FpI>();
    }
    pr = mul(sel4(r, Pd, Rr, p.zz, p.zzz), sel4(r, Pd, Rr, q.zz, q.zzz));
    const FpI PP = quad_from<0>(pr), RR = quad_from<1>(pr), ZZ12 = quad_from<2>(pr), ZZZ12 = quad_from<3>(pr);
    pr = mul(sel4(r, Pd, U1, ZZ12, ZZ12), PP);
    const FpI PPP = quad_from<0>(pr), Q = quad_from<1>(pr), ZZ3 = quad_from<2>(pr);
    const FpI X3 = sub(sub(RR, PPP), dbl(Q));
    pr = mul(sel4(r, Rr, S1, ZZZ12, ZZZ12), sel4(r, sub(Q, X3), PPP, PPP, PPP));
    const FpI t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<FpI>{X3, sub(t0, t1), ZZ3, ZZZ3};
}
// 2P (dbl-2008-s-1); infinity stays infinity (zz = 0 propagates)
__device__ __forceinline__ Xyzz<FpI> g1_dbl4(const Xyzz<FpI> &p, int r, int gb) {
    const FpI U = dbl(p.y);
    FpI pr = mul(sel4(r, U, p.x, U, U), sel4(r, U, p.x, U, U));
    const FpI V = quad_from<0>(pr), XX = quad_from<1>(pr);
    const FpI M = add(dbl(XX), XX);
    pr = mul(sel4(r, U, p.x, M, V), sel4(r, V, V, M, p.zz));
    const FpI W = quad_from<0>(pr), S = quad_from<1>(pr), MM = quad_from<2>(pr), ZZ3 = quad_from<3>(pr);
    const FpI X3 = sub(MM, dbl(S));
    pr = mul(sel4(r, M, W, W, W), sel4(r, sub(S, X3), p.y, p.zzz, p.zzz));
    const FpI t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<FpI>{X3, sub(t0, t1), ZZ3, ZZZ3};
}
// [|z|]p on the 4 lanes of the group
__device__ __forceinline__ Xyzz<FpI> g1_mul_zabs4(const Xyzz<FpI> &p, int r, int gb) {
    const uint64_t z = K_Z_ABS;
    Xyzz<FpI> acc = p;
    for (int i = 62; i >= 0; i--) {
        acc = g1_dbl4(acc, r, gb);
        if ((z >> i) & 1ull) acc = g1_add4(acc, p, r, gb);
    }
    return acc;
}

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
// Lane 0 of a pair decodes and validates P (pad / < p / on curve), lane 1 does Q; both store the
// Montgomery-form point and a "finite and valid" flag.  The walk and membership kernels then start
// from decoded points: they make no out-of-line calls at all (the decode's ~17 products used the
// out-of-line Fp product, whose callee-saved registers were the walk kernels' scratch traffic).
// Error keys: (pair << 4 | stage << 3 | code), stage 0 = G1 (decode, then subgroup), 1 = G2 -- the
// reference's order inside a pair (src/eip2537.c:1036-1053); atomicMin keeps the first.
__global__ void __launch_bounds__(256)
k_pair_decode(const uint32_t *__restrict__ in, uint32_t k, Aff<Fp> *__restrict__ pmont, Aff<Fp2> *__restrict__ qmont,
              uint8_t *__restrict__ flagP, uint8_t *__restrict__ flagQ, unsigned long long *err, CallMap cm) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 1;
    if (i >= k) return;
    if ((t & 1u) == 0) {
        Aff<Fp> P;
        const int st = decode_point<Fp>(P, in + (size_t)i * kPairWords);
        if (st != E_SUCCESS) { report_pair_error(err, cm, i, (unsigned long long)st); P = Aff<Fp>{fp_zero(), fp_zero()}; }
        pmont[i] = P;
        flagP[i] = (st == E_SUCCESS && !is_inf(P)) ? 1 : 0;
    } else {
        Aff<Fp2> Q;
        const int st = decode_point<Fp2>(Q, in + (size_t)i * kPairWords + 32);
        if (st != E_SUCCESS) { report_pair_error(err, cm, i, 8ull | (unsigned long long)st); Q = Aff<Fp2>{fp2_zero(), fp2_zero()}; }
        qmont[i] = Q;
        flagQ[i] = (st == E_SUCCESS && !is_inf(Q)) ? 1 : 0;
    }
}

// EXCL: one wave per SIMD (batches whose walk + membership waves fit the chip's 1024 SIMDs)
template <bool EXCL>
__global__ void __launch_bounds__(64)
k_pair_check_g1(const Aff<Fp> *__restrict__ pmont, const uint8_t *__restrict__ flagP, uint32_t k, unsigned long long *err, CallMap cm) {
    const int lane = threadIdx.x & 63, r = lane & 3, gb = lane & ~3;
    const uint32_t i = blockIdx.x * 16u + (threadIdx.x >> 2);
    if (EXCL) claim_whole_simd();
    if (i >= k) return;                                       // uniform in the group
    if (!flagP[i]) return;                                    // infinity (a member) or already reported by the decode
    const Aff<Fp> pc = pmont[i];
    const Aff<FpI> p{FpI{pc.x}, FpI{pc.y}};
    // phi(P) == -[z^2]P  (curve.h in_g1)
    const Xyzz<FpI> t = g1_mul_zabs4(g1_mul_zabs4(from_affine(p), r, gb), r, gb);
    const Aff<FpI> phi_neg{mul(p.x, FpI{Fp{{K_BETA}}}), neg(p.y)};
    const bool same = !is_inf(t) && eq(mul(phi_neg.x, t.zz), t.x) && eq(mul(phi_neg.y, t.zzz), t.y);
    if (!same && r == 0) report_pair_error(err, cm, i, (unsigned long long)E_NOT_IN_SUBGROUP);
}

// ---- line walk: 8 lanes per pair, split by Fp2 component ---------------------------------------
// A lone wave issues about one VALU instruction every 5-6 cycles, so one pair per lane made the 63-step
// walk a 2400-product serial chain on 64 waves.  A pair therefore owns a group of 8 lanes = 4 lane pairs;
// the independent Fp2 products of a step are dealt over the lane pairs in rounds of four --
//   doubling   [X^2  Y^2  Z^2  YZ]   [B^2  (X+B)^2  E^2  EX]   [E ZZ  Z3 ZZ  E(D-X3)]
//   addition   six rounds,   closing G2 membership test psi(Q) == -[|z|]Q   three rounds
// -- and exchanged with wavefront shuffles.  Lane (p, q) holds only COMPONENT q of every Fp2 value
// (replicated over the four lane pairs p): the linear steps are component-wise, so they run on Fp, and
// only a product needs the partner lane's component of its two operands (one exchange with lane ^ 1);
// lane pair p computes product p of a round by the schoolbook rule (c0 = a0 b0 - a1 b1,
// c1 = a0 b1 + a1 b0: two Fp products per lane).  Round 1 had every lane hold whole Fp2 values and
// repeat every linear step on both components, in 4-, 8- and 16-lane forms picked by batch size: a
// doubling step was ~8 200 instructions (2 350 multiply-adds, ~1 800 carry-chain additions /
// subtractions plus 830 hazard nops, operand selects) with 1 GB of scratch stores per launch at 2^12
// pairs; this form is ~5 500 without spills to memory, and faster than each of the three at every size
// (2^12 pairs: walk 1.43 -> 1.04 ms; 2 pairs: check 1.92 -> 1.77 ms; 2^15 pairs: 12.1 -> 10.5 ms), so it
// is the only one left.
struct TcFp { FpI x, y, z; };                     // this lane's component of the running point, in [0, 2p)
using Walk8c = PairProd8;                         // lanes.h: products of a round, split by Fp2 component
// lane pair `part` stores line coefficient `part` (a0, a1, a4), each lane its component
__device__ __forceinline__ void store_line_part_c(LineRec *dst, int part, const FpI &v, bool contributes, const Walk8c &w) {
    if (w.p == part) {                      // canonical in memory: the product tree computes in [0, p)
        Fp *slot = reinterpret_cast<Fp *>(&dst->a0) + 2 * part + w.q;
        *slot = contributes ? fp_canon(v) : ((part == 0 && w.q == 0) ? fp_one() : fp_zero());
    }
}
__device__ __forceinline__ void miller_dbl_step_c(TcFp &T, const Walk8c &prod, LineRec *dst, bool contributes) {
    Prod4c pr = prod(T.x, T.y, T.z, T.y, T.x, T.y, T.z, T.z);
    const FpI A = pr.r0, B = pr.r1, ZZ = pr.r2, YZ = pr.r3;
    const FpI E = add(dbl(A), A), XB = add(T.x, B);
    pr = prod(B, XB, E, E, B, XB, E, T.x);
    const FpI C = pr.r0, t = pr.r1, F = pr.r2, EX = pr.r3;
    store_line_part_c(dst, 0, sub(EX, dbl(B)), contributes, prod);              // 3X^3 - 2Y^2
    const FpI D = dbl(sub(sub(t, A), C));
    const FpI X3 = sub(F, dbl(D)), Z3 = dbl(YZ);
    const FpI C8 = dbl(dbl(dbl(C)));
    pr = prod(E, Z3, E, E, ZZ, ZZ, sub(D, X3), ZZ);
    store_line_part_c(dst, 1, neg(pr.r0), contributes, prod);                   // -3X^2 Z^2
    store_line_part_c(dst, 2, pr.r1, contributes, prod);                        // 2YZ^3
    T.x = X3;
    T.y = sub(pr.r2, C8);
    T.z = Z3;
}
__device__ __forceinline__ void miller_add_step_c(TcFp &T, const FpI &Qx, const FpI &Qy, const Walk8c &prod, LineRec *dst, bool contributes) {
    Prod4c pr = prod(T.z, T.z, T.z, T.z, T.z, T.z, T.z, T.z);
    const FpI ZZ = pr.r0;
    pr = prod(Qx, ZZ, Qx, Qx, ZZ, T.z, ZZ, ZZ);
    const FpI U2 = pr.r0, ZZZ = pr.r1;
    pr = prod(Qy, Qy, Qy, Qy, ZZZ, ZZZ, ZZZ, ZZZ);
    const FpI S2 = pr.r0;
    const FpI H = sub(U2, T.x), th = sub(S2, T.y);
    pr = prod(H, T.z, th, th, H, H, th, Qx);
    const FpI HH = pr.r0, Z3 = pr.r1, TH2 = pr.r2, thQx = pr.r3;
    pr = prod(HH, T.x, Z3, HH, H, HH, Qy, H);
    const FpI HHH = pr.r0, V = pr.r1, Z3Qy = pr.r2;
    const FpI X3 = sub(sub(TH2, HHH), dbl(V));
    const FpI VX = sub(V, X3);
    pr = prod(th, T.y, th, th, VX, HHH, VX, VX);
    store_line_part_c(dst, 0, sub(thQx, Z3Qy), contributes, prod);
    store_line_part_c(dst, 1, neg(th), contributes, prod);
    store_line_part_c(dst, 2, Z3, contributes, prod);
    T.x = X3;
    T.y = sub(pr.r0, pr.r1);
    T.z = Z3;
}
template <bool EXCL>
__device__ __forceinline__ void pair_walk8c(const Aff<Fp2> *__restrict__ qmont, const uint8_t *__restrict__ flagP,
                                            const uint8_t *__restrict__ flagQ, uint32_t k, LineRec *__restrict__ lines,
                                            unsigned long long *err, Aff<Fp2> *sQ, const CallMap &cm) {
    const int lane = threadIdx.x & 63, sl = lane & 7, gbase = lane & ~7, gi = lane >> 3, q = sl & 1;
    const uint32_t i = blockIdx.x * 8u + (uint32_t)gi;
    bool q_live = false, contributes = false;
    TcFp T;
    if (EXCL) claim_whole_simd();              // batches whose walk + membership waves fit one per SIMD
    if (i < k) {                              // uniform within a lane group
        q_live = flagQ[i] != 0;
        contributes = q_live && flagP[i] != 0;                   // else the pair contributes 1
        const Aff<Fp2> Q = qmont[i];
        if (sl == 0) sQ[gi] = Q;
        T = TcFp{FpI{sel2(q, Q.x.c0, Q.x.c1)}, FpI{sel2(q, Q.y.c0, Q.y.c1)}, FpI{q ? fp_zero() : fp_one()}};
    }
    __syncthreads();
    if (i >= k) return;
    if (!q_live) {                            // Q at infinity (or undecodable: reported by the decode)
        if (sl < 3) {
            const Fp2 v = sl == 0 ? fp2_one() : fp2_zero();
            for (int s = 0; s < kSteps; s++) (&lines[(size_t)s * k + i].a0)[sl] = v;
        }
        return;
    }
    const Walk8c prod(lane, sl, gbase);
    const uint64_t z = K_Z_ABS;
    int s = 0;
    for (int bit = 62; bit >= 0; bit--) {
        miller_dbl_step_c(T, prod, &lines[(size_t)s * k + i], contributes);
        s++;
        if ((z >> bit) & 1ull) {              // 5 of 63 steps
            const FpI Qx{q ? sQ[gi].x.c1 : sQ[gi].x.c0}, Qy{q ? sQ[gi].y.c1 : sQ[gi].y.c0};
            miller_add_step_c(T, Qx, Qy, prod, &lines[(size_t)s * k + i], contributes);
            s++;
        }
    }
    // T = [|z|]Q (Jacobian).  Q in G2  <=>  psi(Q) == [z]Q = -T:  psi(Q).x Z^2 == X  and  -psi(Q).y Z^3 == Y
    const FpI Qx{q ? sQ[gi].x.c1 : sQ[gi].x.c0}, Qy{q ? sQ[gi].y.c1 : sQ[gi].y.c0};
    const FpI cx = q ? neg(Qx) : Qx, cy = q ? neg(Qy) : Qy;                      // conj: component 1 negated
    const FpI kx{q ? Fp{{K_PSI_X_C1}} : Fp{{K_PSI_X_C0}}}, ky{q ? Fp{{K_PSI_Y_C1}} : Fp{{K_PSI_Y_C0}}};
    Prod4c pr = prod(cx, cy, T.z, cx, kx, ky, T.z, kx);
    const FpI px = pr.r0, py = neg(pr.r1), zz = pr.r2;
    pr = prod(px, zz, px, px, zz, T.z, zz, zz);
    const FpI lhs_x = pr.r0, zzz = pr.r1;
    pr = prod(py, py, py, py, zzz, zzz, zzz, zzz);
    int same = (eq(lhs_x, T.x) && eq(pr.r0, T.y)) ? 1 : 0, zzero = is_zero(T.z) ? 1 : 0;
    same &= __shfl(same, lane ^ 1, 64);                                         // both components
    zzero &= __shfl(zzero, lane ^ 1, 64);
    if ((!same || zzero) && sl == 0)
        report_pair_error(err, cm, i, 8ull | (unsigned long long)E_NOT_IN_SUBGROUP);
}

template <bool EXCL>
__global__ void __launch_bounds__(64)
k_pair_lines8(const Aff<Fp2> *__restrict__ qmont, const uint8_t *__restrict__ flagP, const uint8_t *__restrict__ flagQ,
              uint32_t k, LineRec *__restrict__ lines, unsigned long long *err, CallMap cm) {
    __shared__ Aff<Fp2> sQ[8];
    pair_walk8c<EXCL>(qmont, flagP, flagQ, k, lines, err, sQ, cm);
}
// ---- Fp12 spread over a group of 8 lanes ----------------------------------------------------
// In the product trees an Fp12 element g = sum_k g_k w^k (g_k in Fp2, w^6 = xi) lives in one
// 8-lane group: lane `sub` (0..5) holds g_sub, lanes 6 and 7 idle.  A lane then needs ~24
// VGPRs per element instead of 144 (one-lane Fp12 products spilled ~6 KB of scratch and ran
// ~10x slower per field product), and a dense product costs each lane 6 Fp2 products instead of
// 18.  Operands move between lanes with ds_bpermute (wavefront shuffles).
// Tower <-> w-power order:  [c0.a0 c0.a1 c0.a2 c1.a0 c1.a1 c1.a2] = [g0 g2 g4 g1 g3 g5].
__device__ __forceinline__ int tower_slot(int sub) { return (sub & 1) * 3 + (sub >> 1); }

#ifndef EIP_TREE_MUL
#define EIP_TREE_MUL 2
#endif
__device__ __forceinline__ Fp2 tmul(const Fp2 &a, const Fp2 &b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return mul(a, b);            // host pass only parses this
#elif EIP_TREE_MUL == 1
    return fp2_mul_regcall(a, b);
#elif EIP_TREE_MUL == 2
    return fp2_mul_body(a, b);
#else
    return mul(a, b);
#endif
}
// out_k = sum_i a_i b_{k-i}, indices mod 6, times xi when the index wrapped
__device__ __forceinline__ Fp2 grp_mul(const Fp2 &a, const Fp2 &b, int sub, int gbase) {
    Fp2 acc = fp2_zero();
    for (int i = 0; i < 6; i++) {
        int j = sub - i;
        const bool wrap = j < 0;
        if (wrap) j += 6;
        Fp2 ai = shfl_from(a, gbase + i);
        Fp2 bj = shfl_from(b, gbase + (j & 7));
        Fp2 t = tmul(ai, bj);
        Fp2 tx = mul_xi(t);
        acc = add(acc, sel2(wrap ? 0 : 1, tx, t));
    }
    return acc;
}
// Dense product with 24 instead of 36 Fp2 products, three per lane on all 8 lanes.  In tower terms
// g = a0 + a1 w with a0 = (g0, g2, g4), a1 = (g1, g3, g5) in Fp6 = Fp2[v]/(v^3 - xi), v = w^2:
//   lane pair p computes one Fp6 product  (p = 0: a0 b0,  1: a1 b1,  2: a0 b1,  3: a1 b0)
//   by Karatsuba: lane q = 0 the x_i y_i, lane q = 1 the (x_i + x_j)(y_i + y_j); the two lanes swap
//   their products and both form z = x y; then  c0 = a0 b0 + v (a1 b1),  c1 = a0 b1 + a1 b0.
// Out of line on purpose: one copy of the three inlined Fp2 products serves every call site, and a
// block makes only ~10 of these calls (the stack-argument cost that ruled this out for the
// per-line products does not matter here).
#ifndef EIP_TREE_KARATSUBA
#define EIP_TREE_KARATSUBA 1
#endif
template <int CALLER> static __device__ __noinline__ Fp2 grp_mul_k(Fp2 a, Fp2 b, int slot, int gbase) {
    const int p = slot >> 1, q = slot & 1;
    const int px = p & 1, py = (p == 1 || p == 2) ? 1 : 0;
    // round r multiplies x_r y_r (q = 0) or (x_r + x_r')(y_r + y_r') with r' = r + 1 mod 3 (q = 1); the
    // operands are gathered per round (registers: this function must fit beside a second wave)
    Fp2 m[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int r2 = r == 2 ? 0 : r + 1;
        const Fp2 xr = shfl_from(a, gbase + 2 * r + px), xs = shfl_from(a, gbase + 2 * r2 + px);
        const Fp2 yr = shfl_from(b, gbase + 2 * r + py), ys = shfl_from(b, gbase + 2 * r2 + py);
        m[r] = mul(sel2(q, xr, add(xr, xs)), sel2(q, yr, add(yr, ys)));       // three register-argument Fp calls
    }
    const int partner = gbase + (slot ^ 1);
    const Fp2 n0 = shfl_from(m[0], partner), n1 = shfl_from(m[1], partner), n2 = shfl_from(m[2], partner);
    const Fp2 t0 = sel2(q, m[0], n0), t1 = sel2(q, m[1], n1), t2 = sel2(q, m[2], n2);
    const Fp2 s01 = sel2(q, n0, m[0]), s12 = sel2(q, n1, m[1]), s02 = sel2(q, n2, m[2]);
    const Fp2 z0 = add(t0, mul_xi(sub(sub(s12, t1), t2)));
    const Fp2 z1 = add(sub(sub(s01, t0), t1), mul_xi(t2));
    const Fp2 z2 = add(sub(sub(s02, t0), t2), t1);
    // lane q of a pair publishes z_q in out1; z2 travels on its own
    const Fp2 out1 = sel2(q, z0, z1);
    // slot k = 2i (+1): even  c0_i = (a0 b0)_i + (v a1 b1)_i,  v (z0, z1, z2) = (xi z2, z0, z1);  odd  c1_i = (a0 b1)_i + (a1 b0)_i
    const int k = slot < 6 ? slot : 0, i = k >> 1, odd = k & 1;
    const int pa = odd ? 2 : 0, pb = odd ? 3 : 1;
    const int cb = odd ? i : (i == 0 ? 2 : i - 1);
    const Fp2 fa1 = shfl_from(out1, gbase + 2 * pa + (i == 1 ? 1 : 0)), fa2 = shfl_from(z2, gbase + 2 * pa);
    const Fp2 fb1 = shfl_from(out1, gbase + 2 * pb + (cb == 1 ? 1 : 0)), fb2 = shfl_from(z2, gbase + 2 * pb);
    const Fp2 first = sel2(i == 2 ? 0 : 1, fa2, fa1);
    Fp2 second = sel2(cb == 2 ? 0 : 1, fb2, fb1);
    const Fp2 sxi = mul_xi(second);
    if (!odd && i == 0) second = sxi;
    return add(first, second);
}
template <int CALLER> __device__ __forceinline__ Fp2 grp_mul_dense(const Fp2 &a, const Fp2 &b, int sub, int gbase) {
#if EIP_TREE_KARATSUBA && defined(__HIP_DEVICE_COMPILE__)
    return grp_mul_k<CALLER>(a, b, sub, gbase);      // one copy per kernel: it inherits that kernel's register budget
#else
    return grp_mul(a, b, sub, gbase);
#endif
}
// (a1, a4) of a stored line times (xP, yP): four Fp products on lanes 0..3 of the group
__device__ __forceinline__ void scale_line(LineRec &l, const Aff<Fp> &P, int sub, int gbase) {
    const int r = sub & 3;
    const Fp w = sel4(r, l.a1.c0, l.a1.c1, l.a4.c0, l.a4.c1);
#if defined(__HIP_DEVICE_COMPILE__)
    const Fp q = fp_mul_cols(w, sel2(r < 2 ? 0 : 1, P.x, P.y));
#else
    const Fp q = mul(w, r < 2 ? P.x : P.y);
#endif
    l.a1 = Fp2{shfl_from(q, gbase), shfl_from(q, gbase + 1)};
    l.a4 = Fp2{shfl_from(q, gbase + 2), shfl_from(q, gbase + 3)};
}
// f * (a0 + a1 w^2 + a4 w^3)
__device__ __forceinline__ Fp2 grp_mul_line(const Fp2 &f, const LineRec &l, int sub, int gbase) {
    int j2 = sub - 2, j3 = sub - 3;
    const bool w2 = j2 < 0, w3 = j3 < 0;
    if (w2) j2 += 6;
    if (w3) j3 += 6;
    Fp2 f2 = shfl_from(f, gbase + (j2 & 7)), f3 = shfl_from(f, gbase + (j3 & 7));
    Fp2 t0 = tmul(f, l.a0);
    Fp2 t2 = tmul(f2, l.a1);
    Fp2 t3 = tmul(f3, l.a4);
    Fp2 t2x = mul_xi(t2), t3x = mul_xi(t3);
    return add(add(t0, sel2(w2 ? 0 : 1, t2x, t2)), sel2(w3 ? 0 : 1, t3x, t3));
}
// product of the 8 groups of a wave, left in group 0
// (only the first `live` groups hold something other than one: levels whose partners are all one are skipped)
template <int CALLER> __device__ __forceinline__ void wave_group_product(Fp2 &acc, int lane, int sub, int gbase, int live = 8) {
    const int gi = lane >> 3;
    for (int step = 1; step < 8 && step < live; step <<= 1) {
        Fp2 partner = shfl_from(acc, (lane + 8 * step) & 63);
        if ((gi & (2 * step - 1)) == 0) acc = grp_mul_dense<CALLER>(acc, partner, sub, gbase);
    }
}


// grid (blocks, 68 steps), 256 threads = 32 groups: each group folds `group_lines` lines of its
// step into a dense element, then wave tree (shuffles) and an LDS step across the 4 waves.
__global__ void __launch_bounds__(256, 2)     // two blocks per CU: the grid is sized to one such round
k_pair_tree(const LineRec *__restrict__ lines, const Aff<Fp> *__restrict__ pmont, uint32_t k, Fp2 *__restrict__ blk_out,
            uint32_t group_lines) {
    __shared__ Fp2 sm[4][6];
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, gbase = lane & ~7;
    const uint32_t g = blockIdx.x * 32u + (threadIdx.x >> 3);
    // the group's first line seeds the accumulator (a0 + a1 w^2 + a4 w^3 in w-power slots 0, 2, 3)
    Fp2 acc = sub == 0 ? fp2_one() : fp2_zero();
    if ((size_t)g * group_lines < k) {
        LineRec l = lines[(size_t)s * k + (size_t)g * group_lines];
        scale_line(l, pmont[(size_t)g * group_lines], sub, gbase);
        acc = sub == 0 ? l.a0 : sub == 2 ? l.a1 : sub == 3 ? l.a4 : fp2_zero();
    }
    for (uint32_t j = 1; j < group_lines; j++) {
        const uint32_t i = g * group_lines + j;
        if (i < k) {
            LineRec l = lines[(size_t)s * k + i];
            scale_line(l, pmont[i], sub, gbase);
            acc = grp_mul_line(acc, l, sub, gbase);
        }
    }
    // groups of this block / wave that hold lines (small batches leave most of them at one: their
    // tree levels are skipped -- 5 dense products less at k <= 8, the whole chain of a 2-pair check)
    const uint32_t ngroups = (k + group_lines - 1) / group_lines;
    const int live_blk = (int)min(32u, ngroups - min(ngroups, blockIdx.x * 32u));
    const int live_wave = max(0, min(8, live_blk - 8 * wave));
    wave_group_product<0>(acc, lane, sub, gbase, live_wave);
    if (live_blk <= 8) {                       // uniform in the block: wave 0 alone holds the product
        if (wave == 0 && lane < 6) blk_out[((size_t)s * gridDim.x + blockIdx.x) * 6 + tower_slot(lane)] = acc;
        return;
    }
    if (lane < 6) sm[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < 16) {
        // two levels: groups 0 and 1 of wave 0 take (wave 0 x wave 1) and (wave 2 x wave 3), then group 0 joins them
        const int gq = lane >> 3, sidx = sub < 6 ? sub : 0;
        acc = grp_mul_dense<0>(sm[2 * gq][sidx], sm[2 * gq + 1][sidx], sub, gbase);
        const Fp2 partner = shfl_from(acc, 8 + sub);
        if (gq == 0 && live_blk > 16) acc = grp_mul_dense<0>(acc, partner, sub, 0);
        if (lane < 6) blk_out[((size_t)s * gridDim.x + blockIdx.x) * 6 + tower_slot(lane)] = acc;
    }
}

// one wave per step: fold the per-block products (tower-layout Fp12 = 6 Fp2) into L_s
__global__ void __launch_bounds__(64)
k_pair_tree2(const Fp2 *__restrict__ blk_out, uint32_t nblk, Fp2 *__restrict__ step_out) {
    const int s = blockIdx.x;
    const int lane = threadIdx.x, sub = lane & 7, gbase = lane & ~7, gi = lane >> 3;
    // each group's first element seeds its accumulator (lanes 6, 7 of a group carry zeros)
    Fp2 acc = sub == 0 ? fp2_one() : fp2_zero();
    if ((uint32_t)gi < nblk) {
        acc = blk_out[((size_t)s * nblk + gi) * 6 + tower_slot(sub < 6 ? sub : 0)];
        if (sub >= 6) acc = fp2_zero();
    }
    for (uint32_t b = gi + 8; b < nblk; b += 8) {
        Fp2 partner = blk_out[((size_t)s * nblk + b) * 6 + tower_slot(sub < 6 ? sub : 0)];
        acc = grp_mul_dense<1>(acc, partner, sub, gbase);
    }
    wave_group_product<1>(acc, lane, sub, gbase, (int)min(8u, nblk));
    if (lane < 6) step_out[(size_t)s * 6 + tower_slot(lane)] = acc;
}

// Coalesced batch: grid (calls, 68 steps), one block per (call, step).  A call of the batch has at most
// 32 * group_lines pairs, so one block folds all its lines of a step and its output IS that call's L_s
// ([call][step][6] Fp2, tower layout).
__global__ void __launch_bounds__(256, 2)
k_pair_tree_batch(const LineRec *__restrict__ lines, const Aff<Fp> *__restrict__ pmont, uint32_t k, const uint32_t *__restrict__ coff,
                  Fp2 *__restrict__ step_out, uint32_t group_lines) {
    __shared__ Fp2 sm[4][6];
    const int j = blockIdx.x, s = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, gbase = lane & ~7;
    const uint32_t base = coff[j], kj = coff[j + 1] - base;
    const uint32_t g = threadIdx.x >> 3;                       // 32 groups
    Fp2 acc = sub == 0 ? fp2_one() : fp2_zero();
    if (g * group_lines < kj) {
        const uint32_t i = base + g * group_lines;
        LineRec l = lines[(size_t)s * k + i];
        scale_line(l, pmont[i], sub, gbase);
        acc = sub == 0 ? l.a0 : sub == 2 ? l.a1 : sub == 3 ? l.a4 : fp2_zero();
    }
    for (uint32_t t = 1; t < group_lines; t++) {
        const uint32_t li = g * group_lines + t;
        if (li < kj) {
            const uint32_t i = base + li;
            LineRec l = lines[(size_t)s * k + i];
            scale_line(l, pmont[i], sub, gbase);
            acc = grp_mul_line(acc, l, sub, gbase);
        }
    }
    const uint32_t ngroups = (kj + group_lines - 1) / group_lines;
    const int live_blk = (int)min(32u, ngroups);
    const int live_wave = max(0, min(8, live_blk - 8 * wave));
    wave_group_product<2>(acc, lane, sub, gbase, live_wave);
    Fp2 *out = step_out + ((size_t)j * kSteps + s) * 6;
    if (live_blk <= 8) {                       // uniform in the block: wave 0 alone holds the product
        if (wave == 0 && lane < 6) out[tower_slot(lane)] = acc;
        return;
    }
    if (lane < 6) sm[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < 16) {
        const int gq = lane >> 3, sidx = sub < 6 ? sub : 0;
        acc = grp_mul_dense<2>(sm[2 * gq][sidx], sm[2 * gq + 1][sidx], sub, gbase);
        const Fp2 partner = shfl_from(acc, 8 + sub);
        if (gq == 0 && live_blk > 16) acc = grp_mul_dense<2>(acc, partner, sub, 0);
        if (lane < 6) out[tower_slot(lane)] = acc;
    }
}

// Decode, membership and line walk of K pairs that belong to M calls (M = 1: one call, no table): the
// part of the device pipeline a single call and a coalesced batch share.  `d_coff` is the device copy of
// the call offsets (nullptr for M = 1).  Leaves ev_a / ev_b around the walk and the fork joined.
static int pairing_front(Engine *e, const uint32_t *in, size_t k, const uint32_t *d_coff, int M, unsigned long long *err,
                         LineRec *lines, Aff<Fp> *pmont, Aff<Fp2> *qmont, uint8_t *flagP, uint8_t *flagQ) {
    const uint32_t line_blocks = (uint32_t)((k + 7) / 8), check_blocks = (uint32_t)((k + 15) / 16);
    const bool excl = line_blocks + check_blocks <= 1024u;       // one wave per SIMD while everything fits the chip
    {
        LastPlan lp{};
        snprintf(lp.kernel, sizeof lp.kernel, "%s", excl ? "k_pair_lines8<true>" : "k_pair_lines8<false>");
        lp.windows = kSteps; lp.lanes = 8; lp.units = (uint32_t)k;
        e->last_plan = lp;
    }
    const CallMap cm{d_coff, M};
    hipStream_t s = e->stream;
    HIPCHK(hipMemsetAsync(err, 0xFF, (size_t)M * 8, s));
    HIPCHK(hipEventRecord(e->ev_start, s));
    hipLaunchKernelGGL(k_pair_decode, dim3((uint32_t)((2 * k + 255) / 256)), dim3(256), 0, s, in, (uint32_t)k, pmont, qmont, flagP, flagQ, err, cm);
    HIPCHK(hipEventRecord(e->ev_j3, s));
    // fork: the G1 membership kernel runs beside the line walk
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_j3, 0));
    if (excl)
        hipLaunchKernelGGL(k_pair_check_g1<true>, dim3(check_blocks), dim3(64), 0, e->stream2, pmont, flagP, (uint32_t)k, err, cm);
    else
        hipLaunchKernelGGL(k_pair_check_g1<false>, dim3(check_blocks), dim3(64), 0, e->stream2, pmont, flagP, (uint32_t)k, err, cm);
    HIPCHK(hipEventRecord(e->ev_j2, e->stream2));
    HIPCHK(hipEventRecord(e->ev_a, s));
    if (excl) hipLaunchKernelGGL(k_pair_lines8<true>, dim3(line_blocks), dim3(64), 0, s, qmont, flagP, flagQ, (uint32_t)k, lines, err, cm);
    else hipLaunchKernelGGL(k_pair_lines8<false>, dim3(line_blocks), dim3(64), 0, s, qmont, flagP, flagQ, (uint32_t)k, lines, err, cm);
    HIPCHK(hipEventRecord(e->ev_b, s));
    return E_SUCCESS;
}

int pairing_device(Engine *e, const void *d_in, size_t k, uint32_t *ml_words) {
    if (k == 0 || k >= (1ull << 27)) return E_MEMORY_ERROR;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) != 0) {
        fprintf(stderr, "[eip2537_hip] device input must be 4-byte aligned\n");
        return E_MEMORY_ERROR;
    }
    // lines folded serially per 8-lane group: as few as possible while the whole grid (blocks x 68
    // steps) still fits in ONE round of 2 blocks per CU (512 slots); one block more than that and
    // the kernel takes two block-times (measured: 544 blocks 1.8 ms)
    const uint32_t max_blocks_per_step = 512u / kSteps;                         // 7
    const uint32_t group_lines = (uint32_t)std::max<size_t>(1, (k + 32 * max_blocks_per_step - 1) / (32 * max_blocks_per_step));
    const uint32_t tree_blocks = (uint32_t)((k + 32 * group_lines - 1) / (32 * group_lines));
    HIPCHK(e->misc.reserve(64));
    HIPCHK(e->pts.reserve(k * sizeof(Aff<Fp>)));
    HIPCHK(e->digits.reserve(k * sizeof(Aff<Fp2>) + 2 * k + 64));      // decoded Q of every pair, then the two flag arrays
    HIPCHK(e->partial.reserve((size_t)kSteps * k * sizeof(LineRec)));
    HIPCHK(e->winout.reserve(((size_t)kSteps * tree_blocks + kSteps) * sizeof(Fp12)));
    auto *err = reinterpret_cast<unsigned long long *>(e->misc.p);
    auto *lines = reinterpret_cast<LineRec *>(e->partial.p);
    auto *pmont = reinterpret_cast<Aff<Fp> *>(e->pts.p);               // P of every pair, Montgomery form
    auto *qmont = reinterpret_cast<Aff<Fp2> *>(e->digits.p);
    auto *flagP = reinterpret_cast<uint8_t *>(qmont + k), *flagQ = flagP + k;
    auto *blk_out = reinterpret_cast<Fp2 *>(e->winout.p);                 // [step][block] tower-layout Fp12
    auto *step_out = blk_out + (size_t)kSteps * tree_blocks * 6;           // [step] Fp12
    hipStream_t s = e->stream;
    int st = pairing_front(e, reinterpret_cast<const uint32_t *>(d_in), k, nullptr, 1, err, lines, pmont, qmont, flagP, flagQ);
    if (st) return st;
    // one block per step (k <= 32 group_lines): its output IS L_s, same [step][6] layout as step_out
    hipLaunchKernelGGL(k_pair_tree, dim3(tree_blocks, kSteps), dim3(256), 0, s, lines, pmont, (uint32_t)k,
                       tree_blocks == 1 ? step_out : blk_out, group_lines);
    if (tree_blocks > 1) hipLaunchKernelGGL(k_pair_tree2, dim3(kSteps), dim3(64), 0, s, blk_out, tree_blocks, step_out);
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());

    unsigned long long herr = 0;
    std::vector<Fp12> L(kSteps);
    HIPCHK(hipMemcpyAsync(&herr, err, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L.data(), step_out, (size_t)kSteps * sizeof(Fp12), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    if (herr != ~0ull) return (int)(herr & 7ull);
    // (folding the k sparse lines of a step on the host instead of launching the tree was tried for
    // k <= 4: it won while the tree cost 0.42 ms, and lost -- 2.8 vs 2.4 ms at k = 4 -- once it did not)
    const Fp12 F = miller_product_from_steps(L.data());
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
    const uint32_t group_lines = (kmax + 31u) / 32u;
    HIPCHK(e->misc.reserve(64 + (size_t)M * 8 + (size_t)(M + 1) * 4));
    HIPCHK(e->pts.reserve(k * sizeof(Aff<Fp>)));
    HIPCHK(e->digits.reserve(k * sizeof(Aff<Fp2>) + 2 * k + 64));
    HIPCHK(e->partial.reserve((size_t)kSteps * k * sizeof(LineRec)));
    HIPCHK(e->winout.reserve((size_t)M * kSteps * sizeof(Fp12)));
    auto *err = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(e->misc.p) + 64);              // [M]
    auto *d_coff = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 64 + (size_t)M * 8);     // [M + 1]
    auto *lines = reinterpret_cast<LineRec *>(e->partial.p);
    auto *pmont = reinterpret_cast<Aff<Fp> *>(e->pts.p);
    auto *qmont = reinterpret_cast<Aff<Fp2> *>(e->digits.p);
    auto *flagP = reinterpret_cast<uint8_t *>(qmont + k), *flagQ = flagP + k;
    auto *step_out = reinterpret_cast<Fp2 *>(e->winout.p);                // [call][step] Fp12
    hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(d_coff, coff, (size_t)(M + 1) * 4, hipMemcpyHostToDevice, s));
    int st = pairing_front(e, reinterpret_cast<const uint32_t *>(d_in), k, d_coff, M, err, lines, pmont, qmont, flagP, flagQ);
    if (st) return st;
    hipLaunchKernelGGL(k_pair_tree_batch, dim3((uint32_t)M, kSteps), dim3(256), 0, s, lines, pmont, (uint32_t)k, d_coff, step_out, group_lines);
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> herr((size_t)M);
    HIPCHK(hipMemcpyAsync(herr.data(), err, (size_t)M * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L_words, step_out, (size_t)M * kSteps * sizeof(Fp12), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    for (int j = 0; j < M; j++) rc[j] = herr[(size_t)j] != ~0ull ? (int)(herr[(size_t)j] & 7ull) : E_SUCCESS;
    return E_SUCCESS;
}

}  // namespace eip
