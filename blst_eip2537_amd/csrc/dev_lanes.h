// Device executor of the 8-lane groups of limbk.h (host twin: HostLanes8 there): one share per lane, exchanges are DPP moves.
// Used by the pairing line walk (pairing.hip) and by the G2 fold / reduce kernels (msm.hip, g2_limb.h).
#pragma once
#include "lanes.h"
#include "limbk.h"

namespace eip {

template <class Fn> __device__ __forceinline__ FpL map_limbs(const FpL &a, Fn fn) {
    FpL r;
#pragma unroll
    for (int k = 0; k < 13; k++) r.l[k] = fn(a.l[k]);
    return r;
}
struct DevLanes8 {                       // 8 lanes = 4 lane pairs, lane (p, q) holds component q  (line walk)
    int p, q;
    template <int K> __device__ __forceinline__ LV<K, 1> swap(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return quad_perm<kDppSwap>(v); })}};
    }
    template <int J, int K> __device__ __forceinline__ LV<K, 1> from_pair(const LV<K, 1> &a) const {
        return LV<K, 1>{{map_limbs(a.l[0], [](uint32_t v) { return group8_pair<J>(v); })}};
    }
    template <int A, int B> __device__ __forceinline__ LV<max2(A, B), 1> pick_q(const LV<A, 1> &a, const LV<B, 1> &b) const {
        LV<max2(A, B), 1> r;
#pragma unroll
        for (int k = 0; k < 13; k++) r.l[0].l[k] = pick2(q == 0, a.l[0].l[k], b.l[0].l[k]);
        return r;
    }
    template <int A, int B, int C, int D>
    __device__ __forceinline__ LV<max4(A, B, C, D), 1> pick_p(const LV<A, 1> &a, const LV<B, 1> &b, const LV<C, 1> &c, const LV<D, 1> &d) const {
        LV<max4(A, B, C, D), 1> r;
#pragma unroll
        for (int k = 0; k < 13; k++) r.l[0].l[k] = pick4(p, a.l[0].l[k], b.l[0].l[k], c.l[0].l[k], d.l[0].l[k]);
        return r;
    }
    __device__ __forceinline__ LanePred<1> both(const LanePred<1> &m) const {
        const uint32_t v = m.b[0] ? 1u : 0u;
        return LanePred<1>{{(v & quad_perm<kDppSwap>(v)) != 0}};
    }
};

}  // namespace eip
