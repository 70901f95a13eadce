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
// value of lane + off (lanes past the end read their own value; callers mask them out)
template <class T> __device__ __forceinline__ T shfl_down(const T &a, int off) {
    const int lane = threadIdx.x & 63;
    return shfl_from(a, lane + off < 64 ? lane + off : lane);
}

__device__ __forceinline__ Fp sel4(int r, const Fp &a, const Fp &b, const Fp &c, const Fp &d) {
    Fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = r == 0 ? a.l[i] : r == 1 ? b.l[i] : r == 2 ? c.l[i] : d.l[i];
    return o;
}
__device__ __forceinline__ FpI sel4(int r, const FpI &a, const FpI &b, const FpI &c, const FpI &d) { return FpI{sel4(r, a.v, b.v, c.v, d.v)}; }
__device__ __forceinline__ Fp2 sel4(int r, const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) {
    return Fp2{sel4(r, a.c0, b.c0, c.c0, d.c0), sel4(r, a.c1, b.c1, c.c1, d.c1)};
}
__device__ __forceinline__ Fp sel2(int r, const Fp &a, const Fp &b) {
    Fp o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.l[i] = r == 0 ? a.l[i] : b.l[i];
    return o;
}
__device__ __forceinline__ Fp2 sel2(int r, const Fp2 &a, const Fp2 &b) {
    Fp2 o;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        o.c0.l[i] = r == 0 ? a.c0.l[i] : b.c0.l[i];
        o.c1.l[i] = r == 0 ? a.c1.l[i] : b.c1.l[i];
    }
    return o;
}

}  // namespace eip
