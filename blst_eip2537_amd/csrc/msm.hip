// G1 / G2 multi-scalar multiplication on gfx950: Pippenger bucket method.
//
// Replaces the reference's MSM algorithms (naive loop src/eip2537.c:564-616, Bos-Coster heap
// :619-708, G2 clones :851-998).  The output of the ABI is a canonical affine encoding, so any
// correct algorithm is bit-identical to the reference's (SURVEY.md 8a "parity principle").
//
// Pipeline (all kernels on one stream; one thread per unit named in brackets):
//   k_msm_decode   [record]  wire decode + validation + Montgomery form; AoS affine points;
//                            signed c-bit window digits -> digit array [window][record]
//   k_msm_hist / k_msm_slicescan  per-(slice, window) LDS histograms, scanned over slices
//   k_msm_scan     [1 block] exclusive scan of the histogram -> entry offsets, task offsets
//   k_msm_scatter  [slice x window]  counting-sort scatter of (point index, sign) by bucket, ranks by LDS atomics
//   k_msm_tasks    [bucket]  split every bucket into tasks of <= L entries
//   k_msm_task_*   [task]    counting sort of the tasks by length (equal trip counts per wave)
//   k_msm_accum    [task]    XYZZ mixed additions over the task's entries        (dominant)
//   k_msm_fold_*   [split bucket]  sum of the task partials of buckets split into several tasks
//   k_msm_reduce1/4 [segment] running-sum sum_j j*B_j over S buckets + offset multiple, then a
//                            wavefront-shuffle tree and an LDS step -> one point per block;
//                            G2 spreads every point operation over 4 lanes (k_msm_reduce4)
//   host                     adds the few per-block points of each window, Horner over windows
#include <algorithm>
#include <mutex>
#include <stdio.h>
#include <type_traits>
#include <vector>
#include "codec.h"
#include "lanes.h"
#include "limb30.h"
#include "dev_lanes.h"
#include "g2_limb.h"
#include "ifma_horner.h"
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

struct Task { uint32_t start, len; };
static constexpr uint32_t kSlice = 32768;             // largest slice of the counting-sort kernels (records per block)
static constexpr uint32_t kLdsWords = 32768;          // 65536 packed 16-bit counters
// Records per (slice, window) block of the counting-sort kernels.  Round 3 used 32 768 everywhere: a 2^16-record call then ran its
// scatter on 40 blocks (66 of its 143 us of sort stage) and a 245 760-record shard of a staged call its coarse scatter on 128 blocks
// (45 us for a quarter of the records the whole 2^20 sorts in 102).
static uint32_t msm_slice_for(uint32_t n, int W, bool sort2) {
    static const uint32_t env = [] { const char *v = getenv("EIP2537_SORT_SLICE"); return v ? (uint32_t)atoi(v) : 0u; }();     // A/B
    if (env >= 1024u && env <= kSlice && (env & (env - 1u)) == 0u) return env;
    // the largest slice that still gives the W windows ~256 blocks in all (2^20 records at c = 16: 32 768 -> 512 blocks, measured
    // better than 8 192 -> 2 048 blocks: 0.558 against 0.612 ms of sort stage; profiles/r04_sort_stage.txt)
    uint32_t s = kSlice;
    const uint32_t floor_ = sort2 ? 4096u : 2048u;
    while (s > floor_ && (uint64_t)((n + s - 1u) / s) * (uint64_t)W < 256u) s >>= 1;
    return s;
}

ChipShape chip_shape(int device) {
    static std::mutex mu;
    static ChipShape cache[64];
    std::lock_guard<std::mutex> lk(mu);
    const int slot = device >= 0 && device < 64 ? device : 0;
    if (!cache[slot].cus) {
        uint32_t cus = 0;
        if (const char *v = getenv("EIP2537_HIP_CUS")) cus = (uint32_t)atoi(v);
        if (!cus) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = (uint32_t)prop.multiProcessorCount;
            else { (void)hipGetLastError(); cus = 256u; }
        }
        cus = std::min(std::max(cus, 8u), 1024u);
        cache[slot] = ChipShape{cus, 4u * cus};
        if (cus != 256u)
            fprintf(stderr, "[eip2537_hip] note: device %d has %u compute units (launch shapes were tuned on the 256 of a whole MI355X)\n", device, cus);
    }
    return cache[slot];
}

// Window width measured best per size class on MI355X (profiles/r01_window_sweep.txt).  The work
// model below (products per record-window and per bucket) ranks plans well once the kernels are
// throughput-bound (n >= 2^17) but not below, where the call is a sum of serial chains: the
// longest bucket of the accumulate and ~2S + 2 log2(buckets) + 10 point operations of the reduce.
// There the widths whose top window needs no merging (c = 5, 8, 13, 16 leave 6/8/9/16 top bits)
// win by 15-25 % over their neighbours.
static int msm_measured_width(uint32_t n, bool g2) {
    // round 2 re-sweep with the current kernels (profiles/r02_window_sweep.txt): c = 11 takes the band
    // between the small and the mid plans (G1 2^12: 0.83 against 0.90 ms; G2 2^12: 1.64 against 1.93 ms),
    // and G2 keeps c = 13 up to 2^18 records (5.7 against 6.0 ms for the work model's choice)
    // round 4: with the two-level reduces of the c = 8 (G1) and c = 13 plans the c = 11 band (2 049 .. 8 192 records) is gone -- G1 2^12
    // 0.62 -> 0.56 ms, 2^13 0.655 -> 0.574; G2 2^13 1.07 -> 1.03 (profiles/r04_window_sweep.txt, last section)
    if (n <= 2048) return 8;
    if (g2) return n <= (1u << 18) ? 13 : 0;      // larger G2 inputs: not measured, use the model
    return n <= (1u << 17) ? 13 : 16;
}
MsmPlan msm_make_plan(uint32_t n, int c_override, bool g2) {
    if (!c_override) c_override = msm_measured_width(n, g2);
    MsmPlan best{};
    double best_cost = 1e300;
    for (int c = 4; c <= 16; c++) {
        if (c_override > 0 && c != c_override) continue;      // -1: the work model alone
        MsmPlan pl{};
        pl.n = n;
        pl.c = c;
        pl.W = (256 + c - 1) / c;
        pl.topbits = 256 - (pl.W - 1) * c;
        // a top window of only a few bits would put ~n/2 records into each of its buckets (split
        // buckets, same-address atomics): let it absorb the window below instead
        if (pl.topbits < 8 && pl.W > 1) { pl.W -= 1; pl.topbits += c; }
        if (pl.topbits > 16) continue;          // LDS histogram: 65536 packed 16-bit counters
        pl.B = 1u << (c - 1);
        pl.BT = 1u << pl.topbits;
        pl.NB = (uint32_t)(pl.W - 1) * pl.B + pl.BT;
        // mixed add ~10 products per (record, window); reduce ~2 full adds (~14 products) per
        // bucket plus a fixed per-thread tail
        double cost = (double)n * pl.W * 10.0 + (double)pl.NB * 30.0;
        if (cost < best_cost) { best_cost = cost; best = pl; }
    }
    // a forced width with no valid plan (c = 14: the merged top window would need 18 bits) falls
    // back to the planner's own choice rather than returning an empty plan
    if (best_cost == 1e300 && c_override) return msm_make_plan(n, -1, g2);
    MsmPlan &pl = best;
    pl.slice = kSlice;                           // refined in msm_device_t (msm_slice_for)
    pl.L = 64;                                   // refined in msm_device_t
    pl.S = 16;                                   // refined per field in msm_device_t
    pl.max_entries = (uint64_t)n * pl.W;
    pl.max_tasks = (uint32_t)(pl.NB + pl.max_entries / pl.L + 1);
    return pl;
}

// Signed window digits of an unreduced 256-bit scalar.  Windows 0..W-2 are signed with digits in
// [-(B-1)..B]; the top window is unsigned and absorbs the last carry (value <= 2^topbits), so no
// extra carry window exists.  fn(global bucket id, negate, digit != 0) is called for EVERY window on
// every lane, so that the callers' wave-level ballots and shuffles stay converged.
template <class Fn>
__device__ __forceinline__ void for_each_digit(const uint32_t k[8], const MsmPlan &pl, Fn &&fn) {
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = k[i];
    const uint32_t c = (uint32_t)pl.c;
    const uint32_t mask = (1u << c) - 1u;
    uint32_t carry = 0;
    for (int w = 0; w < pl.W - 1; w++) {
        uint32_t d = (s[0] & mask) + carry;
#pragma unroll
        for (int i = 0; i < 7; i++) s[i] = (s[i] >> c) | (s[i + 1] << (32u - c));
        s[7] >>= c;
        uint32_t ng = d > pl.B ? 1u : 0u;
        if (ng) d = (mask + 1u) - d;
        carry = ng;
        fn((uint32_t)w * pl.B + d - 1u, ng, d != 0u);
    }
    uint32_t d = s[0] + carry;
    fn((uint32_t)(pl.W - 1) * pl.B + d - 1u, 0u, d != 0u);
}


// Point record of the limb-form accumulate (limb30.h): x R', y R' and -y R' in 13 limbs each (the sign
// of a window digit then only picks an address), 14 dwords apart so that every coordinate is 8-byte aligned.
struct PtL { uint32_t x[14], y[14], ny[14]; };
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
// Point record of the limb-form G2 accumulate (k_msm_accum2c_l): per Fp2 COMPONENT q the limbs of x_q R', y_q R' and -y_q R' -- the lane
// that holds component q reads c[q] for its own operands and c[q ^ 1] for its partner's (no lane exchange for the loaded point).
struct PtL2 { uint32_t c[2][3][14]; };
// XYZZ point over Fp2 in limb form, as the G2 accumulate leaves it and the G2 fold / reduce kernels read it: per component q the limbs of
// x_q, y_q, zz_q, zzz_q (standard bounds of g2_limb.h); infinity = all limbs 0
struct Pt2L { uint32_t c[2][4][14]; };
__device__ __forceinline__ void store_limb_record(void *, uint32_t, const Aff<Fp> &) {}
__device__ __forceinline__ void store_limb_record(void *recs, uint32_t i, const Aff<Fp2> &a) {
    PtL2 *r = reinterpret_cast<PtL2 *>(recs) + i;
    const FpL y0 = fpl_from_mont(a.y.c0), y1 = fpl_from_mont(a.y.c1);
    store_limbs(r->c[0][0], fpl_from_mont(a.x.c0));
    store_limbs(r->c[0][1], y0);
    store_limbs(r->c[0][2], negL<1>(y0));
    store_limbs(r->c[1][0], fpl_from_mont(a.x.c1));
    store_limbs(r->c[1][1], y1);
    store_limbs(r->c[1][2], negL<1>(y1));
}
// Wire -> limb record in one go (G1, limb-form plans): the coordinates go from raw words straight to x R', y R'
// (one product of the R world each, by R R' mod p) and the curve equation is checked on limbs -- 2 products
// + 2 limb squarings + 1 limb product instead of the 7 products of decode_point() followed by a conversion.
// Same verdicts in the same order as decode_point(): INVALID_ELEMENT (pad bytes, >= p) before NOT_ON_CURVE,
// (0, 0) is infinity and skips the curve test (reference src/eip2537.c:320-343).
__device__ __forceinline__ int fp_decode_raw(Fp &raw, const uint32_t *w) {       // -1 invalid, 0 zero, 1 non-zero
    const Fp p = fp_p();
    const uint32_t pad = w[0] | w[1] | w[2] | w[3];
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        raw.l[11 - k] = bswap32(w[4 + k]);
        nz |= w[4 + k];
    }
    uint32_t borrow = 0;     // raw < p  <=>  raw - p borrows
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t s = (uint64_t)raw.l[i] - p.l[i] - borrow;
        borrow = (uint32_t)(s >> 32) & 1u;
    }
    if (pad != 0 || borrow == 0) return -1;
    return nz != 0;
}
// E_SUCCESS with live = false for infinity; on success with live = true the record is written
__device__ __forceinline__ int decode_point_limbs(PtL *dst, const uint32_t *w, bool &live) {
    Fp xr, yr;
    const int sx = fp_decode_raw(xr, w), sy = fp_decode_raw(yr, w + Wire<Fp>::kCoordWords);
    live = false;
    if (sx < 0 || sy < 0) return E_INVALID_ELEMENT;
    if (sx == 0 && sy == 0) return E_SUCCESS;
    const Fp k{{K_R384_R390_MODP}};
    const Fp xm = fp_mul_cols(xr, k), ym = fp_mul_cols(yr, k);    // x R', y R', canonical (inlined: no out-of-line call in the decode kernels)
    const FpL xl = to_limbs(xm), yl = to_limbs(ym);
    const FpL rhs = addL(mulL(sqrL(xl), xl), FpL{{K_B1_R390_30}});      // x^3 + 4 in the R' world, < 3p
    if (!is_zero_modp(subL<3>(sqrL(yl), rhs), 5)) return E_NOT_ON_CURVE;
    store_limbs(dst->x, xl);
    store_limbs(dst->y, yl);
    store_limbs(dst->ny, to_limbs(neg(ym)));
    live = true;
    return E_SUCCESS;
}

template <class F>
__global__ void __launch_bounds__(256)
k_msm_decode(const uint32_t *__restrict__ in, MsmPlan pl, Aff<F> *__restrict__ pts, void *__restrict__ limb_recs,
             uint32_t *__restrict__ digits, unsigned long long *err, uint32_t err_base) {
    PtL *ptl = std::is_same<F, Fp>::value ? reinterpret_cast<PtL *>(limb_recs) : nullptr;     // G1: PtL; G2: PtL2 (store_limb_record)
    // pl.n records at `in`: a whole call, or one record shard of a staged call -- err_base is then the shard's first record, so
    // that the error word still orders bad records by their index in the CALL
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool live = false;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < pl.n) {
        const uint32_t *w = in + (size_t)i * Wire<F>::kMsmRecWords;
        int st;
        if (std::is_same<F, Fp>::value && ptl) {              // uniform: limb records straight from the wire
            st = decode_point_limbs(&ptl[i], w, live);
        } else {
            Aff<F> a;
            st = decode_point_inl<F>(a, w);
            if (st == E_SUCCESS && !is_inf(a)) {
                if (limb_recs) store_limb_record(limb_recs, i, a);
                else pts[i] = a;
                live = true;
            }
        }
        if (st != E_SUCCESS) atomicMin(err, ((unsigned long long)(err_base + i) << 3) | (unsigned long long)st);
        else if (live) decode_scalar(k, w + Wire<F>::kPointWords);
    }
    if (i >= pl.n) return;
    // digit array, window-major so that one window's digits of consecutive records are contiguous:
    // 0 = no entry, else (bucket value << 1) | negate
    int wi = 0;
    for_each_digit(k, pl, [&](uint32_t g, uint32_t ng, bool nz) {
        const uint32_t v = g - (uint32_t)wi * pl.B + 1u;                // bucket value 1..nb_w
        digits[(size_t)wi * pl.n + i] = (live && nz) ? ((v << 1) | ng) : 0u;
        wi++;
    });
}

// Three-launch exclusive scan over the bucket histogram (<= 1024 x 1024 buckets):
//   k_msm_scan_sums  [1024 buckets per block]  block totals of (entries, tasks)
//   k_msm_scan_top   [1 block]                 exclusive scan of the block totals
//   k_msm_scan_apply [1024 buckets per block]  local scan + block base -> offsets, taskoff
__device__ __forceinline__ void block_scan_1024(uint32_t &e, uint32_t &k, uint32_t *se, uint32_t *st) {
    // inclusive Hillis-Steele scan of (e, k) over the 1024 threads of the block
    const uint32_t t = threadIdx.x;
    se[t] = e;
    st[t] = k;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        uint32_t ve = t >= off ? se[t - off] : 0u, vt = t >= off ? st[t - off] : 0u;
        __syncthreads();
        se[t] += ve;
        st[t] += vt;
        __syncthreads();
    }
    e = se[t];
    k = st[t];
}
// (round 4, measured and reverted: running k_msm_scan_top / k_msm_task_scan / k_sort_window_bases in the LAST block of their producers --
// ticket counter, __threadfence -- saved three 5-us launches and cost 0.15 ms per 2^20-record sort stage: a device-scope fence on this
// multi-XCD part writes the XCD's L2 back, and every block of a 544-block grid paid for one.  profiles/r04_sort_stage.txt)
__global__ void __launch_bounds__(1024)
k_msm_scan_sums(const uint32_t *__restrict__ counts, uint32_t NB, uint32_t lshift, uint32_t *__restrict__ blk) {
    __shared__ uint32_t se[1024], st[1024];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t lm = (1u << lshift) - 1u;
    uint32_t cnt = i < NB ? counts[i] : 0u;
    uint32_t e = cnt, k = (cnt + lm) >> lshift;
    block_scan_1024(e, k, se, st);
    if (threadIdx.x == 1023u) { blk[2 * blockIdx.x] = e; blk[2 * blockIdx.x + 1] = k; }
}
// also clears what k_msm_tasks accumulates into next (round 3: two memsets per call): the split-bucket counters totals[2], [3] and the
// task-length histograms
__global__ void __launch_bounds__(1024)
k_msm_scan_top(uint32_t *__restrict__ blk, uint32_t nblk, uint32_t NB, uint32_t *__restrict__ taskoff, uint32_t *totals, uint32_t *__restrict__ lenhist) {
    __shared__ uint32_t se[1024], st[1024];
    const uint32_t t = threadIdx.x;
    uint32_t e0 = t < nblk ? blk[2 * t] : 0u, k0 = t < nblk ? blk[2 * t + 1] : 0u;
    uint32_t e = e0, k = k0;
    block_scan_1024(e, k, se, st);
    if (t < nblk) { blk[2 * t] = e - e0; blk[2 * t + 1] = k - k0; }     // exclusive bases
    if (t == 1023u) { taskoff[NB] = k; totals[0] = e; totals[1] = k; totals[2] = 0u; totals[3] = 0u; }
    if (t < 4u * 65u + 4u) lenhist[t] = 0u;
}
__global__ void __launch_bounds__(1024)
k_msm_scan_apply(const uint32_t *__restrict__ counts, uint32_t NB, uint32_t lshift, const uint32_t *__restrict__ blk,
                 uint32_t *__restrict__ offsets, uint32_t *__restrict__ taskoff) {
    __shared__ uint32_t se[1024], st[1024];
    const uint32_t i = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t lm = (1u << lshift) - 1u;
    uint32_t cnt = i < NB ? counts[i] : 0u;
    uint32_t e0 = cnt, k0 = (cnt + lm) >> lshift;
    uint32_t e = e0, k = k0;
    block_scan_1024(e, k, se, st);
    if (i < NB) {
        offsets[i] = blk[2 * blockIdx.x] + e - e0;
        taskoff[i] = blk[2 * blockIdx.x + 1] + k - k0;
    }
}

// ---- counting sort of the (window, bucket) digits without global atomics -------------------------
// Scattered global atomics run at ~25 G/s on this part (two passes of 16.8 M cost 1.35 ms at
// 2^20).  Instead the records are cut into slices of kSlice; one block owns one (slice, window),
// keeps that window's whole histogram in LDS as packed 16-bit counters (65536 x 2 B = 128 KB,
// a slice cannot overflow them) and counts / ranks with LDS atomics only:
//   k_msm_hist       [slice x window]  LDS histogram -> hist16[window][slice][bucket]
//   k_msm_slicescan  [bucket]          exclusive scan over the slices -> base[window][slice][bucket]
//                                      and the bucket totals counts[g]
//   k_msm_scatter    [slice x window]  rank inside (slice, bucket) by LDS atomic; position =
//                                      offsets[g] + base + rank

// WORDS: LDS words of the packed counters -- kLdsWords for the 16-bit windows, 4 096 (16 KB: two blocks per CU) while a window
// has at most 8 192 buckets (c <= 13)
template <uint32_t WORDS>
__global__ void __launch_bounds__(1024)
k_msm_hist(const uint32_t *__restrict__ digits, MsmPlan pl, uint32_t nslices, uint32_t nbmax,
           uint32_t *__restrict__ hist16, uint32_t slice0) {
    __shared__ uint32_t h[WORDS];
    const uint32_t slice = slice0 + blockIdx.x, w = blockIdx.y;
    const uint32_t nbw = (w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B;
    const uint32_t words = (nbw + 1u) / 2u;
    for (uint32_t t = threadIdx.x; t < words; t += 1024u) h[t] = 0;
    __syncthreads();
    const uint32_t lo = slice * pl.slice, hi = min(lo + pl.slice, pl.n);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
        const uint32_t v = digits[(size_t)w * pl.n + i];
        if (v) { const uint32_t b = (v >> 1) - 1u; atomicAdd(&h[b >> 1], 1u << (16u * (b & 1u))); }
    }
    __syncthreads();
    uint32_t *dst = hist16 + ((size_t)w * nslices + slice) * (nbmax / 2u);
    for (uint32_t t = threadIdx.x; t < words; t += 1024u) dst[t] = h[t];
}

__global__ void __launch_bounds__(256)
k_msm_slicescan(const uint32_t *__restrict__ hist16, MsmPlan pl, uint32_t nslices, uint32_t nbmax,
                uint32_t *__restrict__ base, uint32_t *__restrict__ counts, const uint32_t *__restrict__ only_if = nullptr) {
    if (only_if && !*only_if) return;
    // one thread per packed pair of buckets of one window
    const uint32_t w = blockIdx.y, pair = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nbw = (w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B;
    if (2u * pair >= nbw) return;
    uint32_t run0 = 0, run1 = 0;
    for (uint32_t sl = 0; sl < nslices; sl++) {
        const size_t row = (size_t)w * nslices + sl;
        const uint32_t v = hist16[row * (nbmax / 2u) + pair];
        base[row * nbmax + 2u * pair] = run0;
        base[row * nbmax + 2u * pair + 1u] = run1;
        run0 += v & 0xffffu;
        run1 += v >> 16;
    }
    counts[(size_t)w * pl.B + 2u * pair] = run0;
    if (2u * pair + 1u < nbw) counts[(size_t)w * pl.B + 2u * pair + 1u] = run1;
}

template <uint32_t WORDS>
__global__ void __launch_bounds__(1024)
k_msm_scatter(const uint32_t *__restrict__ digits, MsmPlan pl, uint32_t nslices, uint32_t nbmax,
              const uint32_t *__restrict__ base, const uint32_t *__restrict__ offsets, uint32_t *__restrict__ entries, uint32_t passes) {
    __shared__ uint32_t h[WORDS];
    // XCD-aware block order (speed only): blocks are dealt round-robin over the 8 XCDs, so block ids
    // with equal id % 8 share an L2.  All slices of a window go to one such group: the window's
    // entries region (4 B x n, 4 MB at 2^20) is then filled from one L2, where the 4-byte stores of
    // the 32 slices into the same 128-byte lines merge, instead of reaching HBM as partial sectors
    // from eight L2s (531 MB written for 67 MB of entries before).
    const uint32_t group = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t slice = j % nslices, w = group + 8u * (j / nslices);
    if (w >= (uint32_t)pl.W) return;
    const uint32_t nbw = (w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B;
    const uint32_t words = (nbw + 1u) / 2u;
    for (uint32_t t = threadIdx.x; t < words; t += 1024u) h[t] = 0;
    __syncthreads();
    const uint32_t *brow = base + ((size_t)w * nslices + slice) * nbmax;
    const uint32_t *orow = offsets + (size_t)w * pl.B;
    const uint32_t lo = slice * pl.slice, hi = min(lo + pl.slice, pl.n);
    // The window's entries region (4 B x n: 4 MB at 2^20) is as large as the XCD's whole L2, so with the
    // streamed digit / offset rows passing through the same cache the 4-byte stores left as partial
    // lines (610 MB written for 67 MB of entries).  The slice is therefore scanned `passes` times, each
    // pass placing only the entries of one contiguous range of buckets: the 32 blocks of the window (one
    // XCD) then work on 1 / passes of the region at a time, which stays resident until its lines are full.
    for (uint32_t pass = 0; pass < passes; pass++) {
        const uint32_t blo = (uint32_t)((uint64_t)nbw * pass / passes), bhi = (uint32_t)((uint64_t)nbw * (pass + 1) / passes);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
            const uint32_t v = digits[(size_t)w * pl.n + i];
            if (v) {
                const uint32_t b = (v >> 1) - 1u, sh = 16u * (b & 1u);
                if (b < blo || b >= bhi) continue;
                const uint32_t rank = (atomicAdd(&h[b >> 1], 1u << sh) >> sh) & 0xffffu;
                entries[orow[b] + brow[b] + rank] = (i << 1) | (v & 1u);
            }
        }
        __syncthreads();
    }
}

// ---- partitioned counting sort (c = 16 plans) ------------------------------------------------------------
// The scatter above places every entry directly: 4-byte stores to random positions of a 4 MB window region, 0.28 ms and
// 530 MB written for 67 MB of entries at 2^20 even in four bucket-range passes, on top of per-bucket slice histograms of
// 67 + 134 MB.  Round 3: two passes that only ever write RUNS.  Bucket b = 128 part + fine.
//   k_sort_coarse_hist     [slice x window]  LDS histogram over the <= 512 partitions of the window
//   k_sort_coarse_scan     [window]          exclusive prefix over (partition, slice) in place, window totals
//   k_sort_window_bases    [1 block]         exclusive prefix of the window totals
//   k_sort_coarse_scatter  [slice x window]  ranks by LDS atomics, the slice's entries staged in LDS in partition order and
//                                            written out as one run per partition: (record << 8 | sign << 7 | fine)
//   k_sort_fine            [partition x window]  the partition's run (~4 096 entries) counted and ranked over its 128
//                                            buckets in LDS, written out in bucket order; per-bucket counts
// Entry positions equal the exclusive scan of the counts in bucket order, so the scan / task kernels run unchanged.
// Degenerate inputs: k_sort_fine gives a partition to ONE block, so a partition that holds most of a window (all scalars equal:
// every record of a window in one bucket) would serialise hundreds of thousands of LDS atomics on one address (2^18 equal
// scalars: +0.8 ms).  k_sort_coarse_scan therefore raises a flag when a partition exceeds four times its window's mean; the
// partitioned kernels after it then do nothing and the direct-scatter kernels, launched behind them with the flag as their
// condition, take the plan (they spread a heavy bucket over its 32 768-record slices).  Ordinary inputs pay three empty launches.
static constexpr uint32_t kFineBits = 7, kFine = 1u << kFineBits, kMaxParts = 512, kHeavyFactor = 4;
__global__ void __launch_bounds__(1024)
k_sort_coarse_hist(const uint32_t *__restrict__ digits, MsmPlan pl, uint32_t nslices, uint32_t *__restrict__ chist, uint32_t *__restrict__ heavy) {
    __shared__ uint32_t h[kMaxParts];
    const uint32_t slice = blockIdx.x, w = blockIdx.y;
    const uint32_t parts = ((w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B) >> kFineBits;
    if (threadIdx.x < kMaxParts) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = slice * pl.slice, hi = min(lo + pl.slice, pl.n);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024u) {
        const uint32_t v = digits[(size_t)w * pl.n + i];
        if (v) atomicAdd(&h[((v >> 1) - 1u) >> kFineBits], 1u);
    }
    __syncthreads();
    if (threadIdx.x < parts) chist[((size_t)w * kMaxParts + threadIdx.x) * nslices + slice] = h[threadIdx.x];
}
__global__ void __launch_bounds__(1024)
k_sort_coarse_scan(uint32_t *__restrict__ chist, MsmPlan pl, uint32_t nslices, uint32_t *__restrict__ wtotal, uint32_t *__restrict__ heavy) {
    __shared__ uint32_t sm[1024];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const uint32_t parts = ((w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B) >> kFineBits;
    const uint32_t E = parts * nslices, chunk = (E + 1023u) / 1024u;
    uint32_t *row = chist + (size_t)w * kMaxParts * nslices;
    const uint32_t a = min(t * chunk, E), b = min(a + chunk, E);
    uint32_t sum = 0;
    for (uint32_t k = a; k < b; k++) sum += row[k];
    sm[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = t >= off ? sm[t - off] : 0u;
        __syncthreads();
        sm[t] += v;
        __syncthreads();
    }
    uint32_t run = sm[t] - sum;                                  // exclusive
    for (uint32_t k = a; k < b; k++) { const uint32_t v = row[k]; row[k] = run; run += v; }
    if (t == 1023u) wtotal[w] = sm[t];
    __threadfence_block();
    __syncthreads();
    const uint32_t total = sm[1023], limit = kHeavyFactor * (total / parts) + 4096u;
    for (uint32_t p = t; p < parts; p += 1024u) {
        const uint32_t size = (p + 1u < parts ? row[(size_t)(p + 1u) * nslices] : total) - row[(size_t)p * nslices];
        if (size > limit) atomicOr(heavy, 1u);
    }
}
__global__ void __launch_bounds__(64)
k_sort_window_bases(const uint32_t *__restrict__ wtotal, int W, uint32_t *__restrict__ wbase) {
    const uint32_t i = threadIdx.x;
    uint32_t v = (int)i < W ? wtotal[i] : 0u, incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)i >= off) incl += o;
    }
    if ((int)i <= W) wbase[i] = incl - v;                         // wbase[W] = all entries
}
// STAGE: entries of a slice staged in LDS (>= pl.slice): 16 384 = 64 KB, two blocks per CU
template <uint32_t STAGE>
__global__ void __launch_bounds__(1024)
k_sort_coarse_scatter(const uint32_t *__restrict__ digits, MsmPlan pl, uint32_t nslices, const uint32_t *__restrict__ chist,
                      const uint32_t *__restrict__ wbase, uint32_t *__restrict__ centries, const uint32_t *__restrict__ heavy) {
    __shared__ uint32_t cnt[kMaxParts], start[kMaxParts + 1], gbase[kMaxParts], stage[STAGE];
    if (*heavy) return;
    const uint32_t slice = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    const uint32_t parts = ((w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B) >> kFineBits;
    const uint32_t lo = slice * pl.slice, hi = min(lo + pl.slice, pl.n);
    // Round 4: this slice's partition counts are not counted again (round 3: a first pass over the slice's digits with 32 768 LDS
    // atomics) -- k_sort_coarse_scan left the exclusive prefix over (partition, slice) in chist, so a count is the difference of two
    // neighbours of the flattened array (the last one: the window's total), and the prefix itself is where the run goes.
    const uint32_t *crow = chist + (size_t)w * kMaxParts * nslices;
    if (t < kMaxParts) {
        uint32_t c = 0, g = 0;
        if (t < parts) {
            const uint32_t k = t * nslices + slice, E = parts * nslices;
            g = crow[k];
            c = (k + 1u < E ? crow[k + 1u] : wbase[w + 1] - wbase[w]) - g;
        }
        gbase[t] = g;
        start[t + 1] = c;
    }
    if (t == 0) start[0] = 0;
    __syncthreads();
    // exclusive prefix of the <= 512 counters (threads 0 .. 511, Hillis-Steele in place through `start`)
    for (uint32_t off = 1; off < kMaxParts; off <<= 1) {
        uint32_t v = 0;
        if (t < kMaxParts && t + 1 > off) v = start[t + 1 - off];
        __syncthreads();
        if (t < kMaxParts) start[t + 1] += v;
        __syncthreads();
    }
    if (t < kMaxParts) cnt[t] = 0;
    __syncthreads();
    for (uint32_t i = lo + t; i < hi; i += 1024u) {
        const uint32_t v = digits[(size_t)w * pl.n + i];
        if (v) {
            const uint32_t b = (v >> 1) - 1u, p = b >> kFineBits;
            const uint32_t rank = atomicAdd(&cnt[p], 1u);
            stage[start[p] + rank] = (i << 8) | ((v & 1u) << 7) | (b & (kFine - 1u));
        }
    }
    __syncthreads();
    const uint32_t total = start[parts], base = wbase[w];
    for (uint32_t k = t; k < total; k += 1024u) {
        uint32_t a = 0, b = parts;                               // partition of staged position k: start[a] <= k < start[a + 1]
        while (b - a > 1u) { const uint32_t mid = (a + b) >> 1; if (start[mid] <= k) a = mid; else b = mid; }
        centries[base + gbase[a] + (k - start[a])] = stage[k];
    }
}
static constexpr uint32_t kFineStage = 16384;                    // entries of a partition staged in LDS (4x the mean at 2^20)
__global__ void __launch_bounds__(512)
k_sort_fine(const uint32_t *__restrict__ centries, MsmPlan pl, uint32_t nslices, const uint32_t *__restrict__ chist,
            const uint32_t *__restrict__ wbase, uint32_t *__restrict__ entries, uint32_t *__restrict__ counts, const uint32_t *__restrict__ heavy) {
    __shared__ uint32_t fh[kFine], fstart[kFine + 1], stage[kFineStage];
    const uint32_t part = blockIdx.x, w = blockIdx.y, t = threadIdx.x;
    const uint32_t parts = ((w == (uint32_t)pl.W - 1u) ? pl.BT : pl.B) >> kFineBits;
    if (part >= parts) return;
    if (*heavy) {               // degenerate input: the host re-runs the call with the direct scatter; until then every bucket is empty
        if (t < kFine) counts[(size_t)w * pl.B + (size_t)part * kFine + t] = 0u;
        return;
    }
    const uint32_t *crow = chist + (size_t)w * kMaxParts * nslices;
    const uint32_t begin = wbase[w] + crow[(size_t)part * nslices];
    const uint32_t end = part + 1u < parts ? wbase[w] + crow[(size_t)(part + 1u) * nslices] : wbase[w + 1];
    const uint32_t count = end - begin;
    if (t < kFine) fh[t] = 0;
    __syncthreads();
    for (uint32_t k = t; k < count; k += 512u) atomicAdd(&fh[centries[begin + k] & (kFine - 1u)], 1u);
    __syncthreads();
    if (t < kFine) { counts[(size_t)w * pl.B + (size_t)part * kFine + t] = fh[t]; fstart[t + 1] = fh[t]; }
    if (t == 0) fstart[0] = 0;
    __syncthreads();
    for (uint32_t off = 1; off < kFine; off <<= 1) {
        uint32_t v = 0;
        if (t < kFine && t + 1 > off) v = fstart[t + 1 - off];
        __syncthreads();
        if (t < kFine) fstart[t + 1] += v;
        __syncthreads();
    }
    if (t < kFine) fh[t] = 0;
    __syncthreads();
    const bool staged = count <= kFineStage;                     // uniform
    for (uint32_t k = t; k < count; k += 512u) {
        const uint32_t c = centries[begin + k], f = c & (kFine - 1u);
        const uint32_t pos = fstart[f] + atomicAdd(&fh[f], 1u), ent = ((c >> 8) << 1) | ((c >> 7) & 1u);
        if (staged) stage[pos] = ent; else entries[begin + pos] = ent;
    }
    if (staged) {
        __syncthreads();
        for (uint32_t k = t; k < count; k += 512u) entries[begin + k] = stage[k];
    }
}

__global__ void __launch_bounds__(64)
k_msm_task_scan(const uint32_t *__restrict__ lenhist, uint32_t *__restrict__ lenoff, uint32_t *__restrict__ ranges) {
    // lane i owns length 64 - i (longest first); exclusive prefix over lanes, the second set behind the first.
    // ranges: [0, n0) = slots of the first set, [n0, n0 + n1) = slots of the second
    const uint32_t i = threadIdx.x;
    uint32_t base = 0;
    for (uint32_t set = 0; set < 2; set++) {
        uint32_t v = lenhist[set * 65u + 64 - i], incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, 64);
            if ((int)i >= off) incl += o;
        }
        lenoff[set * 65u + 64 - i] = base + incl - v;
        if (i == 0) lenoff[set * 65u] = 0;
        const uint32_t total = __shfl(incl, 63, 64);
        if (i == 0) { ranges[2 * set] = base; ranges[2 * set + 1] = base + total; }
        base += total;
    }
}
// Layout of Engine::scan_blk (32-bit words): block totals of the bucket scan | task-length histograms (two sets of 65) | their offsets |
// slot ranges of the two sets | window totals | window bases
static constexpr uint32_t kLenHist = 2048, kLenOff = kLenHist + 130, kRanges = kLenOff + 130, kWTotal = kRanges + 4, kWBase = kWTotal + 64,
                          kScanBlkWords = kWBase + 68;
static constexpr uint32_t kTaskItems = 1024;          // buckets / tasks per block of k_msm_tasks / k_msm_task_perm
__global__ void __launch_bounds__(256)
k_msm_tasks(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
            const uint32_t *__restrict__ taskoff, uint32_t NB, uint32_t lshift, Task *__restrict__ tasks,
            uint32_t *__restrict__ split_small, uint32_t *__restrict__ split_big, uint32_t *split_counts,
            uint32_t *__restrict__ lenhist, uint32_t gshift, uint32_t split_g,
            uint32_t *__restrict__ task_bucket = nullptr, Xyzz<FpL> *__restrict__ bacc = nullptr, uint32_t first_shard = 0u) {
    // bucket accumulators (the c = 16 two-level plans, round 4): task_bucket[t] = bucket << 1 | "first task of its bucket" -- the
    // accumulate adds a bucket's first task onto bacc[bucket], the running sum over the record shards of the call, instead of
    // writing a per-task partial; the first shard marks the buckets it leaves empty as infinity (zz = 0), so bacc needs no memset
    // also the histogram of task length classes ceil(len / 2^gshift) in [1, 64] for the sort below (a
    // separate pass over the task array before: 0.03 ms at 2^20); buckets from split_g on are counted as
    // a second set (their tasks are ordered and accumulated on their own: two-level reduce)
    // a block takes kTaskItems buckets (four per thread): its length classes reach the global counters as one atomic per class
    // and block -- with 256 buckets per block those same-address atomics were most of the kernel's 31 us at 2^20
    __shared__ uint32_t h[130];
    if (threadIdx.x < 130) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = 0; j < kTaskItems / 256u; j++) {
    const uint32_t g = blockIdx.x * kTaskItems + j * 256u + threadIdx.x;
    const uint32_t hs = g >= split_g ? 65u : 0u;
    const uint32_t cnt = g < NB ? counts[g] : 0u;
    if (!cnt && bacc && first_shard && g < NB) bacc[g].zz = fpl_zero();
    if (cnt) {
        const uint32_t t0 = taskoff[g], off = offsets[g];
        const uint32_t L = 1u << lshift, gm = (1u << gshift) - 1u;
        if (task_bucket) {
            const uint32_t nt = (cnt + L - 1u) >> lshift;
            for (uint32_t j = 0; j < nt; j++) task_bucket[t0 + j] = (g << 1) | (j == 0u ? 1u : 0u);
        }
        // buckets split into several tasks are folded back into one partial before the reduce:
        // 2..8 tasks by one thread (k_msm_fold_small), more by one block (k_msm_fold_big)
        if (cnt > 8u * L) split_big[atomicAdd(&split_counts[1], 1u)] = g;
        else if (cnt > L) split_small[atomicAdd(&split_counts[0], 1u)] = g;
        const uint32_t full = cnt >> lshift, rest = cnt & (L - 1u);
        for (uint32_t j = 0; j < full; j++) tasks[t0 + j] = Task{off + (j << lshift), L};
        if (rest) tasks[t0 + full] = Task{off + (full << lshift), rest};
        if (full) atomicAdd(&h[hs + ((L + gm) >> gshift)], full);
        if (rest) atomicAdd(&h[hs + ((rest + gm) >> gshift)], 1u);
    }
    }
    __syncthreads();
    if (threadIdx.x < 130 && h[threadIdx.x]) atomicAdd(&lenhist[threadIdx.x], h[threadIdx.x]);
}


// Tasks sorted by length, longest first: bucket loads are Poisson distributed, so without this
// a wave of k_msm_accum runs for the longest of its 64 tasks (~70 % lane utilisation at 2^20).
// Counting sort on the length class ceil(len / (L/64)) in [1, 64]: per-block LDS histogram -> 64 global counters -> a one-wave
// scan -> per-block range reservation -> permutation.
__global__ void __launch_bounds__(256)
k_msm_task_perm(const Task *__restrict__ tasks, const uint32_t *__restrict__ totals, uint32_t *__restrict__ lenoff,
                uint32_t *__restrict__ perm, uint32_t gshift, const uint32_t *__restrict__ taskoff, uint32_t split_g) {
    __shared__ uint32_t h[130], base[130];
    if (threadIdx.x < 130) h[threadIdx.x] = 0;
    __syncthreads();
    constexpr uint32_t kPer = kTaskItems / 256u;          // tasks per thread: one range reservation per class and block
    uint32_t len[kPer], local[kPer];
    const uint32_t t_split = split_g == 0xffffffffu ? 0xffffffffu : taskoff[split_g];     // task ids follow the bucket order
    const uint32_t ntasks = totals[1];
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t t = blockIdx.x * kTaskItems + j * 256u + threadIdx.x;
        len[j] = 0;
        local[j] = 0;
        if (t < ntasks) {
            len[j] = ((tasks[t].len + (1u << gshift) - 1u) >> gshift) + (t >= t_split ? 65u : 0u);
            local[j] = atomicAdd(&h[len[j]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 130 && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&lenoff[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++)
        if (len[j]) perm[base[len[j]] + local[j]] = blockIdx.x * kTaskItems + j * 256u + threadIdx.x;
}

// product used inside the multi-lane point operations: over Fp2 optionally the fully inlined body
#ifndef EIP_G2_HOT_ACC
#define EIP_G2_HOT_ACC 1
#endif
#ifndef EIP_G2_HOT_RED
#define EIP_G2_HOT_RED 1
#endif
template <int HOT, class T> __device__ __forceinline__ T hmul(const T &a, const T &b) { return mul(a, b); }
#if defined(__HIP_DEVICE_COMPILE__)
template <> __device__ __forceinline__ Fp2 hmul<1, Fp2>(const Fp2 &a, const Fp2 &b) { return fp2_mul_body(a, b); }
template <> __device__ __forceinline__ Fp hmul<2, Fp>(const Fp &a, const Fp &b) { return fp_mul_cols(a, b); }
template <> __device__ __forceinline__ Fp2 hmul<2, Fp2>(const Fp2 &a, const Fp2 &b) { return fp2_mul_body(a, b); }
#endif

// ---- G2 accumulate, 2 lanes per task ----------------------------------------------------------
// Over Fp2 a one-lane mixed addition keeps ~15 Fp2 values live (256 VGPR + 191 AGPR + scratch,
// one wave per SIMD) and ran at a fifth of the G1 kernel's product rate.  Here a task owns two
// lanes: the ten products of madd-2008-s are dealt two per round in five rounds
//   [U2 S2] [PP RR] [PPP Q] [ZZ3 Y1*PPP] [R*(Q-X3) ZZZ3]
// (every lane busy in every round), halving the chain and doubling the waves.
template <class T> __device__ __forceinline__ Xyzz<T> madd2(const Xyzz<T> &p, const Aff<T> &q, int r, int gb) {
    if (is_inf(q)) return p;                                   // uniform in the pair of lanes
    if (is_inf(p)) return Xyzz<T>{q.x, q.y, f_one<T>(), f_one<T>()};
    T pr = hmul<2>(sel2(r, q.x, q.y), sel2(r, p.zz, p.zzz));
    const T U2 = quad_perm<kDppPair0>(pr), S2 = quad_perm<kDppPair1>(pr);
    const T Pd = sub(U2, p.x), Rr = sub(S2, p.y);
    if (is_zero(Pd)) {
        if (is_zero(Rr)) return dbl_affine(q);
        return xyzz_inf<T>();
    }
    pr = hmul<2>(sel2(r, Pd, Rr), sel2(r, Pd, Rr));
    const T PP = quad_perm<kDppPair0>(pr), RR = quad_perm<kDppPair1>(pr);
    pr = hmul<2>(sel2(r, Pd, p.x), PP);
    const T PPP = quad_perm<kDppPair0>(pr), Q = quad_perm<kDppPair1>(pr);
    const T X3 = sub(sub(RR, PPP), dbl(Q));
    pr = hmul<2>(sel2(r, p.zz, p.y), sel2(r, PP, PPP));
    const T ZZ3 = quad_perm<kDppPair0>(pr), t1 = quad_perm<kDppPair1>(pr);
    pr = hmul<2>(sel2(r, Rr, p.zzz), sel2(r, sub(Q, X3), PPP));
    const T t0 = quad_perm<kDppPair0>(pr), ZZZ3 = quad_perm<kDppPair1>(pr);
    return Xyzz<T>{X3, sub(t0, t1), ZZ3, ZZZ3};
}
// G1 for the plans whose accumulate is a chain rather than a stream (c <= 13, up to 2^17 records: the
// longest task, not the product rate, sets its time); G2 takes the component-split form below
template <class F>
__global__ void __launch_bounds__(256)
k_msm_accum2(const Aff<F> *__restrict__ pts, const uint32_t *__restrict__ entries, const Task *__restrict__ tasks,
             const uint32_t *__restrict__ perm, const uint32_t *__restrict__ totals, Xyzz<F> *__restrict__ partial) {
    const int lane = threadIdx.x & 63, r = lane & 1, gb = lane & ~1;
    const uint32_t slot = blockIdx.x * 128u + (threadIdx.x >> 1);
    if (slot >= totals[1]) return;                             // uniform in the pair
    const uint32_t t = perm[slot];
    Task tk = tasks[t];
    Xyzz<F> acc = xyzz_inf<F>();
    for (uint32_t e = 0; e < tk.len; e++) {
        uint32_t ent = entries[tk.start + e];
        Aff<F> p = pts[ent >> 1];
        if (ent & 1u) p.y = neg(p.y);
        acc = madd2(acc, p, r, gb);
    }
    if (r == 0) partial[t] = acc;
}

// ---- G2 accumulate, 2 lanes per task, split by Fp2 component --------------------------------------
// k_msm_accum2 over Fp2 kept whole Fp2 values replicated on both lanes of a task (256 VGPR + 68 AGPR + spills,
// one wave per SIMD, 70 % VALU-active: 1.19 ms at 2^16 records; this form 0.91 ms).  Here lane q of the task holds only COMPONENT q of the running
// sum and of the loaded point -- half the registers, linear steps on Fp -- and a round still multiplies
// two pairs of Fp2 operands, lane r computing product r whole by Karatsuba (3 Fp products): before the
// product each lane sends its component of the OTHER lane's operands, after it the other lane's
// component of its result (three one-element exchanges with lane ^ 1 per round).  Values are FpI: kept in
// [0, 2p).  Same formulas and case analysis as madd2.
struct Pair2c { FpI r0, r1; };                    // this lane's component of the two products of a round
__device__ __forceinline__ Pair2c prod2c(const FpI &a0, const FpI &b0, const FpI &a1, const FpI &b1, int r, int lane) {
    // my product is (a_r, b_r); the partner needs my component of (a_{1-r}, b_{1-r})
    const FpI mine_u = sel2(r, a0, a1), mine_v = sel2(r, b0, b1);
    const FpI other_u = quad_perm<kDppSwap>(sel2(r, a1, a0)), other_v = quad_perm<kDppSwap>(sel2(r, b1, b0));
    // lane r holds component r: (u0, u1) = r ? (other, mine) : (mine, other)
    const FpI u0 = sel2(r, mine_u, other_u), u1 = sel2(r, other_u, mine_u);
    const FpI v0 = sel2(r, mine_v, other_v), v1 = sel2(r, other_v, mine_v);
    // schoolbook with one reduction per component (field.h, fp_mul2_cols30): 2 x 507 multiply-adds and one
    // negation against Karatsuba's 3 x 338 with two additions and three subtractions
    const FpI c0 = mul2(u0, v0, u1, neg(v1)), c1 = mul2(u0, v1, u1, v0);
    // keep my component of my product, fetch my component of the partner's product
    const FpI keep = sel2(r, c0, c1), got = quad_perm<kDppSwap>(sel2(r, c1, c0));
    return Pair2c{sel2(r, keep, got), sel2(r, got, keep)};
}
__device__ __forceinline__ bool both2(bool mine, int) { const uint32_t m = mine ? 1u : 0u; return (m & quad_perm<kDppSwap>(m)) != 0; }
// 2Q for affine Q != infinity (mdbl-2008-s-1), components
__device__ __forceinline__ Xyzz<FpI> dbl_affine2c(const Aff<FpI> &a, int r, int lane) {
    const FpI U = dbl(a.y);
    Pair2c pr = prod2c(U, U, a.x, a.x, r, lane);
    const FpI V = pr.r0, XX = pr.r1;
    const FpI M = add(dbl(XX), XX);
    pr = prod2c(U, V, a.x, V, r, lane);
    const FpI W = pr.r0, S = pr.r1;
    pr = prod2c(M, M, W, a.y, r, lane);
    const FpI X3 = sub(pr.r0, dbl(S)), Wy = pr.r1;
    pr = prod2c(M, sub(S, X3), M, M, r, lane);
    return Xyzz<FpI>{X3, sub(pr.r0, Wy), V, W};
}
__device__ __forceinline__ Xyzz<FpI> madd2c(const Xyzz<FpI> &p, const Aff<FpI> &q, int r, int lane) {
    if (both2(is_zero(q.x) && is_zero(q.y), lane)) return p;                     // uniform in the pair of lanes
    if (both2(is_zero(p.zz), lane)) return Xyzz<FpI>{q.x, q.y, FpI{r ? fp_zero() : fp_one()}, FpI{r ? fp_zero() : fp_one()}};
    Pair2c pr = prod2c(q.x, p.zz, q.y, p.zzz, r, lane);
    const FpI Pd = sub(pr.r0, p.x), Rr = sub(pr.r1, p.y);
    if (both2(is_zero(Pd), lane)) {
        if (both2(is_zero(Rr), lane)) return dbl_affine2c(q, r, lane);
        return xyzz_inf<FpI>();
    }
    pr = prod2c(Pd, Pd, Rr, Rr, r, lane);
    const FpI PP = pr.r0, RR = pr.r1;
    pr = prod2c(Pd, PP, p.x, PP, r, lane);
    const FpI PPP = pr.r0, Q = pr.r1;
    const FpI X3 = sub(sub(RR, PPP), dbl(Q));
    pr = prod2c(p.zz, PP, p.y, PPP, r, lane);
    const FpI ZZ3 = pr.r0, t1 = pr.r1;
    pr = prod2c(Rr, sub(Q, X3), p.zzz, PPP, r, lane);
    return Xyzz<FpI>{X3, sub(pr.r0, t1), ZZ3, pr.r1};
}
__global__ void __launch_bounds__(256)
k_msm_accum2c(const Aff<Fp2> *__restrict__ pts, const uint32_t *__restrict__ entries, const Task *__restrict__ tasks,
              const uint32_t *__restrict__ perm, const uint32_t *__restrict__ totals, Xyzz<Fp2> *__restrict__ partial) {
    const int lane = threadIdx.x & 63, r = lane & 1;
    const uint32_t slot = blockIdx.x * 128u + (threadIdx.x >> 1);
    if (slot >= totals[1]) return;                             // uniform in the pair
    const uint32_t t = perm[slot];
    const Task tk = tasks[t];
    Xyzz<FpI> acc = xyzz_inf<FpI>();
    for (uint32_t e = 0; e < tk.len; e++) {
        const uint32_t ent = entries[tk.start + e];
        const Fp *pc = reinterpret_cast<const Fp *>(&pts[ent >> 1]);            // x.c0 x.c1 y.c0 y.c1
        Aff<FpI> p{FpI{pc[r]}, FpI{pc[2 + r]}};
        if (ent & 1u) p.y = neg(p.y);
        acc = madd2c(acc, p, r, lane);
    }
    store_component(&partial[t], acc, r);
}

// field type the accumulate loop computes in: Fp -> FpI (inlined products), Fp2 unchanged
template <class F> struct AccumField { using T = F; };
template <> struct AccumField<Fp> { using T = FpI; };
// FpI values live in [0, 2p) (field.h): whatever leaves the device for the host's canonical arithmetic
// -- the per-block window sums -- goes through canon().  Partial sums that stay on the device (one
// XYZZ point per task) are stored as they are; every kernel that reads them computes in FpI too.
__device__ __forceinline__ Xyzz<FpI> canon(const Xyzz<FpI> &p) {
    return Xyzz<FpI>{FpI{fp_canon(p.x)}, FpI{fp_canon(p.y)}, FpI{fp_canon(p.zz)}, FpI{fp_canon(p.zzz)}};
}
__device__ __forceinline__ const Xyzz<Fp2> &canon(const Xyzz<Fp2> &p) { return p; }
__device__ __forceinline__ const Xyzz<Fp> &canon(const Xyzz<Fp> &p) { return p; }

template <class F>
__global__ void __launch_bounds__(256)
k_msm_accum(const Aff<F> *__restrict__ pts_, const uint32_t *__restrict__ entries,
            const Task *__restrict__ tasks, const uint32_t *__restrict__ perm,
            const uint32_t *__restrict__ totals, Xyzz<F> *__restrict__ partial_) {
    using T = typename AccumField<F>::T;
    static_assert(sizeof(Aff<T>) == sizeof(Aff<F>) && sizeof(Xyzz<T>) == sizeof(Xyzz<F>), "layout");
    const Aff<T> *__restrict__ pts = reinterpret_cast<const Aff<T> *>(pts_);
    Xyzz<T> *__restrict__ partial = reinterpret_cast<Xyzz<T> *>(partial_);
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= totals[1]) return;
    const uint32_t t = perm[slot];
    Task tk = tasks[t];
    Xyzz<T> acc = xyzz_inf<T>();
    for (uint32_t e = 0; e < tk.len; e++) {
        uint32_t ent = entries[tk.start + e];
        Aff<T> p = pts[ent >> 1];
        if (ent & 1u) p.y = neg(p.y);
        acc = madd(acc, p);
    }
    partial[t] = acc;
}

// ---- limb-form accumulate (one lane per task, G1, the throughput-bound plans) ------------------------
// The same mixed addition as madd() on FpL values (limb30.h): no re-slicing between products, no
// conditional corrections.  Bounds held by the accumulator between entries (in units of p):
//   x < 8, y < 4, zz < 2, zzz < 2;   inside: P < 10, R < 6, every product of a b < 630.
__global__ void __launch_bounds__(256)
k_msm_accum_l(const PtL *__restrict__ pts, const uint32_t *__restrict__ entries, const Task *__restrict__ tasks,
              const uint32_t *__restrict__ perm, const uint32_t *__restrict__ totals, Xyzz<Fp> *__restrict__ partial_,
              const uint32_t *__restrict__ range, const uint32_t *__restrict__ task_bucket = nullptr, Xyzz<FpL> *__restrict__ bacc = nullptr,
              uint32_t first_shard = 1u) {
    Xyzz<FpL> *__restrict__ partial = reinterpret_cast<Xyzz<FpL> *>(partial_);     // read by the <Fp, FpL> fold and reduce kernels
    // range: the slots [range[0], range[1]) of the task order (two-level reduce: the top window's upper half first); null: all
    const uint32_t slot = (range ? range[0] : 0u) + blockIdx.x * 256u + threadIdx.x;
    if (slot >= (range ? range[1] : totals[1])) return;
    const uint32_t t = perm[slot];
    const Task tk = tasks[t];
    AccL acc;
    bool inf = true;
    // bucket accumulators: a bucket's first task continues the bucket's running sum over the earlier record shards of the call
    // (stored points keep the standard bounds, which are the loop's own invariants) and leaves its result there
    Xyzz<FpL> *dst = &partial[t];
    if (task_bucket) {
        const uint32_t tb = task_bucket[t];
        if (tb & 1u) {
            dst = &bacc[tb >> 1];
            if (!first_shard) {
                const Xyzz<FpL> b0 = *dst;
                if (!is_zero(b0.zz)) { acc = AccL{b0.x, b0.y, b0.zz, b0.zzz}; inf = false; }
            }
        }
    }
    for (uint32_t e = 0; e < tk.len; e++) {
        const uint32_t ent = entries[tk.start + e];
        const PtL *q = &pts[ent >> 1];
        madd_l(acc, inf, load_limbs(q->x), load_limbs((ent & 1u) ? q->ny : q->y));
    }
    *dst = inf ? xyzz_inf<FpL>() : Xyzz<FpL>{acc.x, acc.y, acc.zz, acc.zzz};
}

// ---- fold: buckets that were split into several tasks ------------------------------------------
// A bucket with more than L entries (the top window when it has only a few bits, duplicate-heavy
// or adversarial inputs: all scalars equal puts every record of a window into ONE bucket) leaves
// several task partials.  They are summed into the bucket's first task slot, the only one the
// reduce kernels read.  (Summing them serially inside the reduce cost 93 ms at n = 2^18, where the
// top window has 1 bit and its two buckets ~2000 tasks each; a block per split bucket cost 30 ms
// when most buckets had 2-4 tasks -- hence two tiers, and L grows with the mean bucket load.)
__device__ __forceinline__ FpL shfl_from(const FpL &a, int src) {
    FpL r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = __shfl(a.l[i], src, 64);
    return r;
}
__device__ __forceinline__ FpL sel4(int r, const FpL &a, const FpL &b, const FpL &c, const FpL &d) {
    FpL o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = pick4(r, a.l[i], b.l[i], c.l[i], d.l[i]);
    return o;
}
__device__ __forceinline__ FpL sel2(int r, const FpL &a, const FpL &b) {
    FpL o;
#pragma unroll
    for (int i = 0; i < 13; i++) o.l[i] = pick2(r == 0, a.l[i], b.l[i]);
    return o;
}
template <int CTRL> __device__ __forceinline__ FpL quad_perm(const FpL &a) {
    FpL r;
#pragma unroll
    for (int i = 0; i < 13; i++) r.l[i] = quad_perm<CTRL>(a.l[i]);
    return r;
}
template <int J> __device__ __forceinline__ FpL quad_from(const FpL &a) { return quad_perm<J * 0x55>(a); }

// ---- G2 accumulate in limb form: 2 lanes per task, lane q holds COMPONENT q of everything -----------------
// k_msm_accum2c computes in FpI (12 x 32-bit words re-sliced into 30-bit limbs around every product, conditional corrections in every
// linear step) and has lane r multiply PRODUCT r of a round whole, which costs three exchanges and a dozen selects per round.  Here values
// stay in limbs (limb30.h: no re-slicing, differences as a + K p - b) and lane q computes component q of EVERY product:
//     (u v)_0 = u0 v0 - u1 v1        (u v)_1 = u0 v1 + u1 v0          one two-product sum with one reduction each (mul2L)
//     (u^2)_0 = (u0 + u1)(u0 - u1)   (u^2)_1 = 2 u0 u1                one product each
//     (a b - c d)_q                                                   one four-product sum (mul4L): Y3 = R (Q - X3) - Y1 PPP
// so a lane needs its partner's component of every operand (one 13-limb exchange with lane ^ 1 per operand, 9 per entry; the loaded
// point's are read from memory) and two selects per product.  4 563 multiply-adds per entry and lane against 5 070.
// Bounds as in madd_l (units of p): x < 8, y < 4, zz, zzz < 2 between entries; P < 10, R < 6; every product sum < 630.
struct C2 { FpL m, o; };                            // my component, my partner's
__device__ __forceinline__ C2 with_partner(const FpL &m) { return C2{m, quad_perm<kDppSwap>(m)}; }
// component q of u v;  KV bounds v
template <int KV> __device__ __forceinline__ FpL mul_c(const C2 &u, const C2 &v, int q) {
    return mul2L(u.m, sel2(q, v.m, v.o), u.o, sel2(q, negL<KV>(v.o), v.m));
}
// component q of u^2;  KU bounds u
template <int KU> __device__ __forceinline__ FpL sqr_c(const C2 &u, int q) {
    return mulL(addL(u.m, sel2(q, u.o, u.m)), sel2(q, subL<KU>(u.m, u.o), u.o));
}
// component q of a b - c d;  KB, KD bound b and d
template <int KB, int KD> __device__ __forceinline__ FpL mul_sub_c(const C2 &a, const C2 &b, const C2 &c, const C2 &d, int q) {
    const FpL nbo = negL<KB>(b.o), ndm = negL<KD>(d.m), ndo = negL<KD>(d.o);
    return mul4L(a.m, sel2(q, b.m, b.o), a.o, sel2(q, nbo, b.m), c.m, sel2(q, ndm, ndo), c.o, sel2(q, d.o, ndm));
}
struct AccL2 { FpL x, y, zz, zzz; };                // one component of an XYZZ point over Fp2
// 2Q for affine Q != infinity (mdbl-2008-s-1); rare
__device__ __forceinline__ AccL2 dbl_affine2c_l(const C2 &qx, const C2 &qy, int q) {
    const C2 U = with_partner(addL(qy.m, qy.m));                             // < 4
    const C2 V = with_partner(sqr_c<4>(U, q));                                // < 2
    const FpL W = mul_c<2>(U, V, q), S = mul_c<2>(qx, V, q), XX = sqr_c<2>(qx, q);
    const C2 M = with_partner(dbl_addL(XX, XX));                             // 3 x^2 < 6
    const FpL X3 = sub2L<4>(sqr_c<6>(M, q), S);                               // < 6
    const FpL Y3 = mul_sub_c<8, 2>(M, with_partner(subL<6>(S, X3)), with_partner(W), qy, q);     // M (S - X3) - W y: < 2
    return AccL2{X3, Y3, V.m, W};
}
// Register budget (measured, profiles/r04_g2_limb_accum.txt): with every partner copy kept only as long as its products the kernel
// needs 236 VGPRs -> two waves per SIMD, 2^18 records 2.49 ms; the first form (276 VGPRs, one wave) 2.65; one wave with the next
// entry's point prefetched 2.75; k_msm_accum2c 2.95.
struct PtC2 { C2 x, y; };
__device__ __forceinline__ PtC2 load_pt2(const PtL2 *__restrict__ pts, uint32_t ent, int q) {
    const PtL2 *p = &pts[ent >> 1];
    const int ys = (ent & 1u) ? 2 : 1;
    return PtC2{C2{load_limbs(p->c[q][0]), load_limbs(p->c[q ^ 1][0])}, C2{load_limbs(p->c[q][ys]), load_limbs(p->c[q ^ 1][ys])}};
}
__global__ void __launch_bounds__(256, 2)
k_msm_accum2c_l(const PtL2 *__restrict__ pts, const uint32_t *__restrict__ entries, const Task *__restrict__ tasks,
                const uint32_t *__restrict__ perm, const uint32_t *__restrict__ totals, Xyzz<Fp2> *__restrict__ partial) {
    const int lane = threadIdx.x & 63, q = lane & 1;
    const uint32_t slot = blockIdx.x * 128u + (threadIdx.x >> 1);
    if (slot >= totals[1]) return;                             // uniform in the pair
    const uint32_t t = perm[slot];
    const Task tk = tasks[t];
    AccL2 acc;
    bool inf = true;                                           // uniform in the pair
#pragma unroll 1
    for (uint32_t e = 0; e < tk.len; e++) {
        const PtC2 pt = load_pt2(pts, entries[tk.start + e], q);
        if (inf) {
            const FpL one_q = q ? fpl_zero() : fpl_one();
            acc = AccL2{pt.x.m, pt.y.m, one_q, one_q};
            inf = false;
            continue;
        }
        const FpL P = subL<8>(mul_c<2>(pt.x, with_partner(acc.zz), q), acc.x);      // < 10
        const FpL R = subL<4>(mul_c<2>(pt.y, with_partner(acc.zzz), q), acc.y);     // < 6
        if (both2(is_zero_modp(P, 10), lane)) {
            if (both2(is_zero_modp(R, 6), lane)) acc = dbl_affine2c_l(pt.x, pt.y, q);
            else inf = true;
            continue;
        }
        // every operand's partner copy lives only as long as its products: zz' right behind PP, zzz' right behind PPP
        const C2 cP = with_partner(P);
        const C2 PP = with_partner(sqr_c<10>(cP, q));
        acc.zz = mul_c<2>(with_partner(acc.zz), PP, q);
        const C2 PPP = with_partner(mul_c<2>(cP, PP, q));
        const FpL Q = mul_c<2>(with_partner(acc.x), PP, q);
        acc.zzz = mul_c<2>(with_partner(acc.zzz), PPP, q);
        const C2 cR = with_partner(R);
        const FpL X3 = sub2L<4>(subL<2>(sqr_c<6>(cR, q), PPP.m), Q);           // R^2 - PPP - 2 Q + 6 p < 8
        acc.y = mul_sub_c<10, 2>(cR, with_partner(subL<8>(Q, X3)), with_partner(acc.y), PPP, q);     // 2 x 6 x 10 + 2 x 4 x 2 < 630; < 2
        acc.x = X3;
    }
    // the task's sum stays in limbs: the G2 fold and reduce kernels compute in limb form too (g2_limb.h)
    Pt2L *dst = reinterpret_cast<Pt2L *>(partial) + t;
    const FpL z = fpl_zero();
    store_limbs(dst->c[q][0], inf ? z : acc.x);
    store_limbs(dst->c[q][1], inf ? z : acc.y);
    store_limbs(dst->c[q][2], inf ? z : acc.zz);
    store_limbs(dst->c[q][3], inf ? z : acc.zzz);
}

// ---- limb-form versions of the lane-split point operations (chain-bound G1 plans, c <= 13) -----------
// Same rounds as add4 / dbl4 / madd2 below and above; values are FpL with the standard bounds of limb30.h
// (x < 8, y < 4, zz, zzz < 2 in units of p), differences are a + K p - b.
// P + Q (add-2008-s), operands replicated on the 4 lanes of the group, complete
__device__ __forceinline__ Xyzz<FpL> dbl4(const Xyzz<FpL> &p, int r, int gb);
__device__ __forceinline__ Xyzz<FpL> add4(const Xyzz<FpL> &p, const Xyzz<FpL> &q, int r, int gb) {
    if (is_zero(q.zz)) return p;                               // uniform in the group
    if (is_zero(p.zz)) return q;
    FpL pr = mulL(sel4(r, p.x, q.x, p.y, q.y), sel4(r, q.zz, p.zz, q.zzz, p.zzz));        // 16, 16, 8, 8 < 630
    const FpL U1 = quad_from<0>(pr), U2 = quad_from<1>(pr), S1 = quad_from<2>(pr), S2 = quad_from<3>(pr);
    const FpL Pd = subL<2>(U2, U1), Rr = subL<2>(S2, S1);       // < 4
    if (is_zero_modp(Pd, 4)) {                                  // same x: double or cancel (rare)
        if (is_zero_modp(Rr, 4)) return dbl4(p, r, gb);
        return xyzz_inf<FpL>();
    }
    pr = mulL(sel4(r, Pd, Rr, p.zz, p.zzz), sel4(r, Pd, Rr, q.zz, q.zzz));
    const FpL PP = quad_from<0>(pr), RR = quad_from<1>(pr), ZZ12 = quad_from<2>(pr), ZZZ12 = quad_from<3>(pr);
    pr = mulL(sel4(r, Pd, U1, ZZ12, ZZ12), PP);
    const FpL PPP = quad_from<0>(pr), Q = quad_from<1>(pr), ZZ3 = quad_from<2>(pr);
    const FpL X3 = sub2L<4>(subL<2>(RR, PPP), Q);               // < 8
    pr = mulL(sel4(r, Rr, S1, ZZZ12, ZZZ12), sel4(r, subL<8>(Q, X3), PPP, PPP, PPP));      // 4 x 10, 2 x 2
    const FpL t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<FpL>{X3, subL<2>(t0, t1), ZZ3, ZZZ3};
}
// 2P (dbl-2008-s-1); infinity stays infinity (zz = 0 propagates)
__device__ __forceinline__ Xyzz<FpL> dbl4(const Xyzz<FpL> &p, int r, int gb) {
    const FpL U = addL(p.y, p.y);                              // < 8
    FpL pr = mulL(sel4(r, U, p.x, U, U), sel4(r, U, p.x, U, U));                           // 64, 64
    const FpL V = quad_from<0>(pr), XX = quad_from<1>(pr);
    const FpL M = dbl_addL(XX, XX);                            // < 6
    pr = mulL(sel4(r, U, p.x, M, V), sel4(r, V, V, M, p.zz));  // 16, 16, 36, 4
    const FpL W = quad_from<0>(pr), S = quad_from<1>(pr), MM = quad_from<2>(pr), ZZ3 = quad_from<3>(pr);
    const FpL X3 = sub2L<4>(MM, S);                            // < 6
    pr = mulL(sel4(r, M, W, W, W), sel4(r, subL<6>(S, X3), p.y, p.zzz, p.zzz));            // 6 x 8, 2 x 4, 2 x 2
    const FpL t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<FpL>{X3, subL<2>(t0, t1), ZZ3, ZZZ3};
}
// acc += (qx, qy) on the two lanes of a task: the rounds of madd2, [U2 S2] [PP RR] [PPP Q] [ZZ3 Y1*PPP] [R*(Q-X3) ZZZ3]
__device__ __forceinline__ void madd2_l(AccL &acc, bool &inf, const FpL &qx, const FpL &qy, int r) {
    if (inf) {                                                 // uniform in the pair of lanes
        acc = AccL{qx, qy, fpl_one(), fpl_one()};
        inf = false;
        return;
    }
    FpL pr = mulL(sel2(r, qx, qy), sel2(r, acc.zz, acc.zzz));
    const FpL U2 = quad_perm<kDppPair0>(pr), S2 = quad_perm<kDppPair1>(pr);
    const FpL Pd = subL<8>(U2, acc.x), Rr = subL<4>(S2, acc.y);                            // < 10, < 6
    if (is_zero_modp(Pd, 10)) {
        if (is_zero_modp(Rr, 6)) acc = dbl_affine_l(qx, qy);
        else inf = true;
        return;
    }
    pr = mulL(sel2(r, Pd, Rr), sel2(r, Pd, Rr));
    const FpL PP = quad_perm<kDppPair0>(pr), RR = quad_perm<kDppPair1>(pr);
    pr = mulL(sel2(r, Pd, acc.x), PP);
    const FpL PPP = quad_perm<kDppPair0>(pr), Q = quad_perm<kDppPair1>(pr);
    const FpL X3 = sub2L<4>(subL<2>(RR, PPP), Q);               // < 8
    pr = mulL(sel2(r, acc.zz, acc.y), sel2(r, PP, PPP));
    const FpL ZZ3 = quad_perm<kDppPair0>(pr), t1 = quad_perm<kDppPair1>(pr);
    pr = mulL(sel2(r, Rr, acc.zzz), sel2(r, subL<8>(Q, X3), PPP));                         // 6 x 10, 2 x 2
    const FpL t0 = quad_perm<kDppPair0>(pr), ZZZ3 = quad_perm<kDppPair1>(pr);
    acc = AccL{X3, subL<2>(t0, t1), ZZ3, ZZZ3};
}
// two lanes per task (the chain-bound G1 plans), limb form: the counterpart of k_msm_accum2<Fp>
__global__ void __launch_bounds__(256)
k_msm_accum2_l(const PtL *__restrict__ pts, const uint32_t *__restrict__ entries, const Task *__restrict__ tasks,
               const uint32_t *__restrict__ perm, const uint32_t *__restrict__ totals, Xyzz<Fp> *__restrict__ partial_) {
    Xyzz<FpL> *__restrict__ partial = reinterpret_cast<Xyzz<FpL> *>(partial_);
    const int lane = threadIdx.x & 63, r = lane & 1;
    const uint32_t slot = blockIdx.x * 128u + (threadIdx.x >> 1);
    if (slot >= totals[1]) return;                             // uniform in the pair
    const uint32_t t = perm[slot];
    const Task tk = tasks[t];
    AccL acc;
    bool inf = true;
    for (uint32_t e = 0; e < tk.len; e++) {
        const uint32_t ent = entries[tk.start + e];
        const PtL *q = &pts[ent >> 1];
        madd2_l(acc, inf, load_limbs(q->x), load_limbs((ent & 1u) ? q->ny : q->y), r);
    }
    if (r == 0) partial[t] = inf ? xyzz_inf<FpL>() : Xyzz<FpL>{acc.x, acc.y, acc.zz, acc.zzz};
}
// per-block window sums leave the device in the canonical form the host reads, whatever the kernel computed in
template <class F, class T> __device__ __forceinline__ void store_canon(Xyzz<F> *out, const Xyzz<T> &v) { *reinterpret_cast<Xyzz<T> *>(out) = canon(v); }
template <> __device__ __forceinline__ void store_canon<Fp, FpL>(Xyzz<Fp> *out, const Xyzz<FpL> &v) { *out = canon(v); }
// out-of-line complete point operations (fold and one-lane reduce kernels)
template <class T> static __device__ __noinline__ void xyzz_add_o(Xyzz<T> *r, const Xyzz<T> *a, const Xyzz<T> *b) { *r = add(*a, *b); }
template <class T> static __device__ __noinline__ void xyzz_dbl_o(Xyzz<T> *r, const Xyzz<T> *a) { *r = dbl(*a); }
// The one-lane reduce takes the limb-form operations inline: through the out-of-line form every operand
// travels by pointer, i.e. through scratch memory (848 B per lane, 670 MB written per launch at 2^20).
template <class T> __device__ __forceinline__ void pt_add(Xyzz<T> &r, const Xyzz<T> &a, const Xyzz<T> &b) { xyzz_add_o<T>(&r, &a, &b); }
template <class T> __device__ __forceinline__ void pt_dbl(Xyzz<T> &r, const Xyzz<T> &a) { xyzz_dbl_o<T>(&r, &a); }
__device__ __forceinline__ void pt_add(Xyzz<FpL> &r, const Xyzz<FpL> &a, const Xyzz<FpL> &b) { r = add(a, b); }
__device__ __forceinline__ void pt_dbl(Xyzz<FpL> &r, const Xyzz<FpL> &a) { r = dbl(a); }
// 2..8 tasks: one thread per bucket
template <class F, class T = typename AccumField<F>::T>    // T: the form the accumulate kernel wrote the partials in (G1: FpI or FpL)
__global__ void __launch_bounds__(256)
k_msm_fold_small(Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
                 const uint32_t *__restrict__ split_counts, uint32_t g_lo = 0u, uint32_t g_hi = 0xffffffffu, Xyzz<T> *__restrict__ bacc = nullptr) {
    Xyzz<T> *__restrict__ partial = reinterpret_cast<Xyzz<T> *>(partial_);
    const uint32_t n = split_counts[0];
    for (uint32_t h = blockIdx.x * 256u + threadIdx.x; h < n; h += gridDim.x * 256u) {
        const uint32_t g = list[h];
        if (g < g_lo || g >= g_hi) continue;                   // two-level reduce: the buckets of one accumulate launch only
        const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
        Xyzz<T> *first = bacc ? &bacc[g] : &partial[t0];       // bucket accumulators: the first task's sum lives in bacc[g]
        Xyzz<T> acc = *first;
        for (uint32_t t = t0 + 1; t < t1; t++) {
            Xyzz<T> pt = partial[t];
            pt_add(acc, acc, pt);                      // inline for limb-form points, out of line otherwise
        }
        *first = acc;
    }
}
// more than 8 tasks: one block per bucket -- 256 strided serial chains, shuffle tree, LDS step
template <class F, class T = typename AccumField<F>::T>
__global__ void __launch_bounds__(256)
k_msm_fold_big(Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
               const uint32_t *__restrict__ split_counts, uint32_t g_lo = 0u, uint32_t g_hi = 0xffffffffu, Xyzz<T> *__restrict__ bacc = nullptr) {
    Xyzz<T> *__restrict__ partial = reinterpret_cast<Xyzz<T> *>(partial_);
    __shared__ Xyzz<T> sm[4];
    const uint32_t nh = split_counts[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t g = list[h];
        if (g < g_lo || g >= g_hi) continue;                   // uniform in the block
        const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
        Xyzz<T> acc = xyzz_inf<T>();
        for (uint32_t t = t0 + threadIdx.x; t < t1; t += 256u) {
            Xyzz<T> pt = (bacc && t == t0) ? bacc[g] : partial[t];
            pt_add(acc, acc, pt);                      // inline for limb-form points, out of line otherwise
        }
        for (int off = 32; off >= 1; off >>= 1) {
            Xyzz<T> o = shfl_down(acc, off);
            if (lane < off) pt_add(acc, acc, o);
        }
        if (lane == 0) sm[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; k++) pt_add(acc, acc, sm[k]);
            if (bacc) bacc[g] = acc; else partial[t0] = acc;
        }
        __syncthreads();
    }
}

// ---- bucket reduce, 4 lanes per running sum -------------------------------------------------
// The reduce is a serial chain of ~60 point operations per segment, so it is bound by the
// latency of one operation, not by throughput.  Every logical thread is therefore a 4-lane
// group that holds its operands replicated; the independent field products of an XYZZ addition
// (4 rounds of <= 4) or doubling (3 rounds) are dealt one per lane and exchanged with
// wavefront shuffles, cutting the chain latency ~3x.
// P + Q (add-2008-s), operands replicated on the 4 lanes of the group, complete
template <class T> __device__ __forceinline__ Xyzz<T> add4(const Xyzz<T> &p, const Xyzz<T> &q, int r, int gb) {
    const bool pinf = is_inf(p), qinf = is_inf(q);            // uniform in the group
    if (qinf) return p;
    if (pinf) return q;
    T pr = hmul<EIP_G2_HOT_RED>(sel4(r, p.x, q.x, p.y, q.y), sel4(r, q.zz, p.zz, q.zzz, p.zzz));
    const T U1 = quad_from<0>(pr), U2 = quad_from<1>(pr), S1 = quad_from<2>(pr), S2 = quad_from<3>(pr);
    const T Pd = sub(U2, U1), Rr = sub(S2, S1);
    if (is_zero(Pd)) {                                         // same x: double or cancel (rare)
        if (is_zero(Rr)) return dbl(p);
        return xyzz_inf<T>();
    }
    pr = hmul<EIP_G2_HOT_RED>(sel4(r, Pd, Rr, p.zz, p.zzz), sel4(r, Pd, Rr, q.zz, q.zzz));
    const T PP = quad_from<0>(pr), RR = quad_from<1>(pr), ZZ12 = quad_from<2>(pr), ZZZ12 = quad_from<3>(pr);
    pr = hmul<EIP_G2_HOT_RED>(sel4(r, Pd, U1, ZZ12, ZZ12), PP);
    const T PPP = quad_from<0>(pr), Q = quad_from<1>(pr), ZZ3 = quad_from<2>(pr);
    const T X3 = sub(sub(RR, PPP), dbl(Q));
    pr = hmul<EIP_G2_HOT_RED>(sel4(r, Rr, S1, ZZZ12, ZZZ12), sel4(r, sub(Q, X3), PPP, PPP, PPP));
    const T t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<T>{X3, sub(t0, t1), ZZ3, ZZZ3};
}
// 2P (dbl-2008-s-1); infinity stays infinity
template <class T> __device__ __forceinline__ Xyzz<T> dbl4(const Xyzz<T> &p, int r, int gb) {
    const T U = dbl(p.y);
    T pr = hmul<EIP_G2_HOT_RED>(sel4(r, U, p.x, U, U), sel4(r, U, p.x, U, U));
    const T V = quad_from<0>(pr), XX = quad_from<1>(pr);
    const T M = add(dbl(XX), XX);
    pr = hmul<EIP_G2_HOT_RED>(sel4(r, U, p.x, M, V), sel4(r, V, V, M, p.zz));
    const T W = quad_from<0>(pr), S = quad_from<1>(pr), MM = quad_from<2>(pr), ZZ3 = quad_from<3>(pr);
    const T X3 = sub(MM, dbl(S));
    pr = hmul<EIP_G2_HOT_RED>(sel4(r, M, W, W, W), sel4(r, sub(S, X3), p.y, p.zzz, p.zzz));
    const T t0 = quad_from<0>(pr), t1 = quad_from<1>(pr), ZZZ3 = quad_from<2>(pr);
    return Xyzz<T>{X3, sub(t0, t1), ZZ3, ZZZ3};
}
template <class T> __device__ __forceinline__ Xyzz<T> small_mul4(const Xyzz<T> &p, uint32_t m, int r, int gb) {
    if (m == 0) return xyzz_inf<T>();
    Xyzz<T> acc = p;                                 // the top bit: no doubling of the point at infinity (three rounds of products on zeros)
    for (int i = 30 - __builtin_clz(m); i >= 0; i--) {
        acc = dbl4(acc, r, gb);
        if ((m >> i) & 1u) acc = add4(acc, p, r, gb);
    }
    return acc;
}

// fold of lightly split buckets with 4-lane additions (G2: the one-lane chain of 1-7 Fp2 additions
// on a few hundred lanes cost 0.19 ms at 2^16, where the 9-bit top window holds 128 records per bucket)
// Round 4: SIXTEEN lanes per bucket -- four 4-lane groups take the partials j, j + 4 and a two-level tree joins them: at most three
// additions in a row instead of seven (the c = 13 plans' 9-bit top window holds n / 512 records per bucket, i.e. eight 16-entry tasks
// at 2^16 records: 45 us of that call were this chain).
template <class F, class T = typename AccumField<F>::T>    // T: the form the accumulate kernel wrote the partials in
__global__ void __launch_bounds__(256)
k_msm_fold_small4(Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
                  const uint32_t *__restrict__ split_counts) {
    Xyzz<T> *__restrict__ partial = reinterpret_cast<Xyzz<T> *>(partial_);
    const uint32_t n = split_counts[0];
    const int lane = threadIdx.x & 63, r = lane & 3, gb = lane & ~3, grp = (lane >> 2) & 3;
    for (uint32_t h = blockIdx.x * 16u + (threadIdx.x >> 4); h < n; h += gridDim.x * 16u) {   // uniform in the 16 lanes of a bucket
        const uint32_t g = list[h], t0 = taskoff[g], t1 = taskoff[g + 1];
        Xyzz<T> acc = xyzz_inf<T>();
        for (uint32_t t = t0 + (uint32_t)grp; t < t1; t += 4u) acc = add4(acc, partial[t], r, gb);      // uniform in the group (<= 2 partials)
        for (int off = 4; off < 16; off <<= 1) {
            const Xyzz<T> o = shfl_from(acc, (lane & ~15) | ((lane + off) & 15));
            if ((lane & (2 * off - 1)) < 4) acc = add4(acc, o, r, gb);
        }
        if ((lane & 15) == 0) partial[t0] = acc;
    }
}

// Reduce grids are one-dimensional and hold only working blocks: `bn` blocks for each of the W - 1 signed
// windows, then the top window's (it can have twice the buckets; a (blocks, W) grid sized for it left
// half of the other windows' blocks idle -- and an idle block still takes a SIMD from a working one).
// Block b's output is winout[b]; the host knows the same map.
struct ReduceGrid { uint32_t bn, bt; };
__device__ __forceinline__ void reduce_block_to_window(const ReduceGrid &rg, const MsmPlan &pl, int &w, uint32_t &bx) {
    const uint32_t b = blockIdx.x, body = (uint32_t)(pl.W - 1) * rg.bn;
    if (b < body) { w = (int)(b / rg.bn); bx = b - (uint32_t)w * rg.bn; }
    else { w = pl.W - 1; bx = b - body; }
}
// 256 threads = 64 four-lane groups, one segment of S buckets per group
template <class F, class T = typename AccumField<F>::T>
__global__ void __launch_bounds__(256, 1)      // latency-bound chain: registers over occupancy
k_msm_reduce4(const Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, MsmPlan pl, ReduceGrid rg,
             Xyzz<F> *__restrict__ winout) {
    const Xyzz<T> *__restrict__ partial = reinterpret_cast<const Xyzz<T> *>(partial_);
    int w;
    uint32_t bx;
    reduce_block_to_window(rg, pl, w, bx);
    claim_whole_simd();                            // one wave per SIMD (lanes.h)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 3, gb = lane & ~3;
    const uint32_t nbw = (w == pl.W - 1) ? pl.BT : pl.B;
    const uint32_t seg = bx * 64u + (threadIdx.x >> 2);
    const uint32_t lo = seg * pl.S;
    Xyzz<T> C = xyzz_inf<T>();
    if (lo < nbw) {                                            // uniform in the group
        const uint32_t hi = min(lo + pl.S, nbw);
        Xyzz<T> R = xyzz_inf<T>(), Q = xyzz_inf<T>();
        for (uint32_t v = hi; v > lo; v--) {
            const uint32_t g = (uint32_t)w * pl.B + v - 1u;
            const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
            if (t1 > t0) R = add4(R, partial[t0], r, gb);       // multi-task buckets were folded into slot t0
            Q = add4(Q, R, r, gb);
        }
        // sum_{v in (lo, hi]} v * B_v = Q + lo * R
        C = add4(Q, small_mul4(R, lo, r, gb), r, gb);
    }
    // product tree over the 16 groups of the wave, then the 4 waves through LDS
    for (int off = 4; off < 64; off <<= 1) {
        Xyzz<T> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 4) C = add4(C, o, r, gb);
    }
    __shared__ Xyzz<T> sm[4];
    if (lane == 0) sm[wave] = C;
    __syncthreads();
    if (wave == 0 && lane < 4) {
        for (int k = 1; k < 4; k++) C = add4(C, sm[k], r, 0);
        if (lane == 0) store_canon<F, T>(&winout[blockIdx.x], C);
    }
}

// fold of lightly split G2 buckets with the 8-lane component-split addition (lanes.h)
__global__ void __launch_bounds__(256)
k_msm_fold_small8c(Xyzz<Fp2> *__restrict__ partial, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
                   const uint32_t *__restrict__ split_counts) {
    const uint32_t n = split_counts[0];
    const int lane = threadIdx.x & 63, sl = lane & 7, gb = lane & ~7, q = sl & 1;
    const PairProd8 prod(lane, sl, gb);
    for (uint32_t h = blockIdx.x * 32u + (threadIdx.x >> 3); h < n; h += gridDim.x * 32u) {   // uniform in the group
        const uint32_t g = list[h], t0 = taskoff[g], t1 = taskoff[g + 1];
        Xyzz<FpI> acc = component_of(partial[t0], q);
        for (uint32_t t = t0 + 1; t < t1; t++) acc = add8c(acc, component_of(partial[t], q), prod);
        if (sl < 2) store_component(&partial[t0], acc, q);
    }
}

// ---- G2 bucket reduce, 8 lanes per running sum, split by Fp2 component -----------------------------------
// k_msm_reduce4<Fp2> kept whole Fp2 points replicated on 4 lanes: R, Q and the operand are 288 dwords per
// lane, so the kernel lived in scratch (2 176 B per lane; 2.7 GB of scratch traffic per launch at 2^16
// records, 1.3 ms).  Here a running sum owns 8 lanes and lane (p, q) holds component q only (lanes.h,
// PairProd8): 144 dwords, products by the schoolbook rule on lane pairs, linear steps on Fp.
// 256 threads = 32 eight-lane groups, one segment of S buckets per group.
__global__ void __launch_bounds__(256, 1)
k_msm_reduce8c(const Xyzz<Fp2> *__restrict__ partial, const uint32_t *__restrict__ taskoff, MsmPlan pl, ReduceGrid rg,
               Xyzz<Fp2> *__restrict__ winout) {
    int w;
    uint32_t bx;
    reduce_block_to_window(rg, pl, w, bx);
    claim_whole_simd();                            // one wave per SIMD (lanes.h)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 7, gb = lane & ~7, q = sl & 1;
    const PairProd8 prod(lane, sl, gb);
    const uint32_t nbw = (w == pl.W - 1) ? pl.BT : pl.B;
    const uint32_t seg = bx * 32u + (threadIdx.x >> 3);
    const uint32_t lo = seg * pl.S;
    const Xyzz<FpI> inf = xyzz_inf<FpI>();
    Xyzz<FpI> C = inf;
    if (lo < nbw) {                                            // uniform in the group
        const uint32_t hi = min(lo + pl.S, nbw);
        Xyzz<FpI> R = inf, Q = inf;
        for (uint32_t v = hi; v > lo; v--) {
            const uint32_t g = (uint32_t)w * pl.B + v - 1u;
            const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
            if (t1 > t0) R = add8c(R, component_of(partial[t0], q), prod);      // multi-task buckets were folded into slot t0
            Q = add8c(Q, R, prod);
        }
        C = add8c(Q, small_mul8c(R, lo, prod), prod);          // sum_{v in (lo, hi]} v * B_v = Q + lo * R
    }
    // tree over the 8 groups of the wave, then the 4 waves through LDS
    for (int off = 8; off < 64; off <<= 1) {
        Xyzz<FpI> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 8) C = add8c(C, o, prod);
    }
    __shared__ Xyzz<FpI> sm[4][2];                  // [wave][component]
    if (lane < 2) sm[wave][lane] = C;
    __syncthreads();
    if (wave == 0 && lane < 8) {
        for (int k = 1; k < 4; k++) C = add8c(C, sm[k][q], prod);
        if (lane < 2) store_component(&winout[blockIdx.x], C, q);
    }
}

// The blocks of a window leave one sum each (<= 12 per window at 2^16 records, 232 in all); adding them on the host cost 0.26 ms
// of a 1.96 ms call (a host G2 addition is ~1.1 us), so one more wave per window does it here: group j of 8 lanes takes the
// blocks j, j + 8, ..., then the tree over the 8 groups -- 4 to 5 additions deep.  winsum[w] = the window's whole sum.
__global__ void __launch_bounds__(64, 1)
k_msm_window_sum8c(const Xyzz<Fp2> *__restrict__ winout, MsmPlan pl, ReduceGrid rg, Xyzz<Fp2> *__restrict__ winsum) {
    const int w = blockIdx.x, lane = threadIdx.x & 63, sl = lane & 7, gb = lane & ~7, q = sl & 1;
    claim_whole_simd();
    const PairProd8 prod(lane, sl, gb);
    const uint32_t nb = w == pl.W - 1 ? rg.bt : rg.bn, b0 = (uint32_t)w * rg.bn;
    Xyzz<FpI> C = xyzz_inf<FpI>();
    for (uint32_t b = (uint32_t)(lane >> 3); b < nb; b += 8u) C = add8c(C, component_of(winout[b0 + b], q), prod);      // uniform in the group
    for (int off = 8; off < 64; off <<= 1) {
        Xyzz<FpI> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 8) C = add8c(C, o, prod);
    }
    if (lane < 2) store_component(&winsum[w], C, q);
}

// ---- G2 fold / bucket reduce in limb form (g2_limb.h) ------------------------------------------------------
// The same kernels as k_msm_fold_small8c / k_msm_reduce8c / k_msm_window_sum8c on the limb-form points that k_msm_accum2c_l leaves
// (Pt2L): no re-slicing around the products, no conditional corrections, and no conversion between the accumulate and the reduce.
// k_msm_fold_big8c_l replaces k_msm_fold_big<Fp2> (whole Fp2 values on every lane: 351 VGPRs + 944 B of scratch) on the path of
// the heavy buckets.
__device__ __forceinline__ XyzzK<1> load_pt2l(const Pt2L *p, int q) {
    XyzzK<1> r;
    r.x.l[0] = load_limbs(p->c[q][0]); r.y.l[0] = load_limbs(p->c[q][1]); r.zz.l[0] = load_limbs(p->c[q][2]); r.zzz.l[0] = load_limbs(p->c[q][3]);
    return r;
}
__device__ __forceinline__ void store_pt2l(Pt2L *p, const XyzzK<1> &v, int q) {
    store_limbs(p->c[q][0], v.x.l[0]); store_limbs(p->c[q][1], v.y.l[0]); store_limbs(p->c[q][2], v.zz.l[0]); store_limbs(p->c[q][3], v.zzz.l[0]);
}
__device__ __forceinline__ XyzzK<1> shfl_from(const XyzzK<1> &a, int src) {
    XyzzK<1> r;
    r.x.l[0] = shfl_from(a.x.l[0], src); r.y.l[0] = shfl_from(a.y.l[0], src); r.zz.l[0] = shfl_from(a.zz.l[0], src); r.zzz.l[0] = shfl_from(a.zzz.l[0], src);
    return r;
}
// sum over the 8 groups of a wave (result in group 0), then over the block's 4 waves through LDS (result in the first group of wave 0)
struct BlockSum8k { uint32_t c[4][2][4][14]; };
__device__ __forceinline__ XyzzK<1> wave_sum8k(const DevLanes8 &x, XyzzK<1> C, int lane) {
    for (int off = 8; off < 64; off <<= 1) {
        const XyzzK<1> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 8) C = add8k(x, C, o);
    }
    return C;
}
__device__ __forceinline__ XyzzK<1> block_sum8k(const DevLanes8 &x, XyzzK<1> C, BlockSum8k &sm, int lane, int wave) {
    C = wave_sum8k(x, C, lane);
    if (lane < 2) store_pt2l(reinterpret_cast<Pt2L *>(sm.c[wave]), C, x.q);
    __syncthreads();
    if (wave == 0 && lane < 8)
        for (int k = 1; k < 4; k++) C = add8k(x, C, load_pt2l(reinterpret_cast<const Pt2L *>(sm.c[k]), x.q));
    return C;
}
__global__ void __launch_bounds__(256)
k_msm_fold_small8c_l(Pt2L *__restrict__ partial, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
                     const uint32_t *__restrict__ split_counts) {
    const uint32_t n = split_counts[0];
    const int sl = threadIdx.x & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    for (uint32_t h = blockIdx.x * 32u + (threadIdx.x >> 3); h < n; h += gridDim.x * 32u) {   // uniform in the group
        const uint32_t g = list[h], t0 = taskoff[g], t1 = taskoff[g + 1];
        XyzzK<1> acc = load_pt2l(&partial[t0], x.q);
        for (uint32_t t = t0 + 1; t < t1; t++) acc = add8k(x, acc, load_pt2l(&partial[t], x.q));
        if (sl < 2) store_pt2l(&partial[t0], acc, x.q);
    }
}
// more than 8 tasks: one block per bucket -- 32 strided chains of 8-lane additions, shuffle tree, LDS step
__global__ void __launch_bounds__(256)
k_msm_fold_big8c_l(Pt2L *__restrict__ partial, const uint32_t *__restrict__ taskoff, const uint32_t *__restrict__ list,
                   const uint32_t *__restrict__ split_counts) {
    __shared__ BlockSum8k sm;
    const uint32_t nh = split_counts[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t g = list[h], t0 = taskoff[g], t1 = taskoff[g + 1];
        XyzzK<1> acc = xyzzk_inf<1>();
        for (uint32_t t = t0 + (threadIdx.x >> 3); t < t1; t += 32u) acc = add8k(x, acc, load_pt2l(&partial[t], x.q));     // uniform in the group
        acc = block_sum8k(x, acc, sm, lane, wave);
        __syncthreads();                                       // every wave's read of partial[t0] and of sm is behind us
        if (threadIdx.x < 2) store_pt2l(&partial[t0], acc, x.q);
    }
}
// bucket reduce: a running sum per 8-lane group over a segment of S buckets (k_msm_reduce8c's map of blocks to windows)
#ifndef EIP_G2L_RED_WAVES
#define EIP_G2L_RED_WAVES 1
#endif
__global__ void __launch_bounds__(256, EIP_G2L_RED_WAVES)
k_msm_reduce8c_l(const Pt2L *__restrict__ partial, const uint32_t *__restrict__ taskoff, MsmPlan pl, ReduceGrid rg, Pt2L *__restrict__ blkout) {
    __shared__ BlockSum8k sm;
    int w;
    uint32_t bx;
    reduce_block_to_window(rg, pl, w, bx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    const uint32_t nbw = (w == pl.W - 1) ? pl.BT : pl.B;
    const uint32_t seg = bx * 32u + (threadIdx.x >> 3);
    const uint32_t lo = seg * pl.S;
    XyzzK<1> C = xyzzk_inf<1>();
    if (lo < nbw) {                                            // uniform in the group
        const uint32_t hi = min(lo + pl.S, nbw);
        XyzzK<1> R = xyzzk_inf<1>(), Q = xyzzk_inf<1>();
        for (uint32_t v = hi; v > lo; v--) {
            const uint32_t g = (uint32_t)w * pl.B + v - 1u;
            const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
            if (t1 > t0) R = add8k(x, R, load_pt2l(&partial[t0], x.q));       // multi-task buckets were folded into slot t0
            Q = add8k(x, Q, R);
        }
        C = add8k(x, Q, small_mul8k(x, R, lo));                // sum_{v in (lo, hi]} v * B_v = Q + lo * R
    }
    C = block_sum8k(x, C, sm, lane, wave);
    if (threadIdx.x < 2) store_pt2l(&blkout[blockIdx.x], C, x.q);
}
// one wave per window adds the window's block sums and leaves the canonical point the host reads
__global__ void __launch_bounds__(64)
k_msm_window_sum8c_l(const Pt2L *__restrict__ blkout, MsmPlan pl, ReduceGrid rg, Xyzz<Fp2> *__restrict__ winsum) {
    const int w = blockIdx.x, lane = threadIdx.x & 63, sl = lane & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    const uint32_t nb = w == pl.W - 1 ? rg.bt : rg.bn, b0 = (uint32_t)w * rg.bn;
    XyzzK<1> C = xyzzk_inf<1>();
    for (uint32_t b = (uint32_t)(lane >> 3); b < nb; b += 8u) C = add8k(x, C, load_pt2l(&blkout[b0 + b], x.q));      // uniform in the group
    C = wave_sum8k(x, C, lane);
    if (lane < 2) {
        const Xyzz<FpI> out{to_fpi(C.x.l[0]), to_fpi(C.y.l[0]), to_fpi(C.zz.l[0]), to_fpi(C.zzz.l[0])};
        store_component(&winsum[w], out, x.q);
    }
}

// ---- bucket reduce, one lane per running sum (used for G1) ------------------------------------
// Measured at 2^20 / c = 16: 1.55 ms against 1.8-2.1 ms for the 4-lane form above (whose per-round
// select / shuffle / stack traffic costs more than the Fp product it parallelises); over Fp2 the
// products are 3x heavier and the 4-lane form wins 2.2 ms to 7.0 ms at 2^16.
template <class F, class T = typename AccumField<F>::T>
__global__ void __launch_bounds__(256, 1)      // one wave per SIMD anyway (claim_whole_simd): the whole register file
k_msm_reduce1(const Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, MsmPlan pl, ReduceGrid rg,
              Xyzz<F> *__restrict__ winout) {
    const Xyzz<T> *__restrict__ partial = reinterpret_cast<const Xyzz<T> *>(partial_);
    int w;
    uint32_t bx;
    reduce_block_to_window(rg, pl, w, bx);
    claim_whole_simd();                            // one wave per SIMD (lanes.h)
    const uint32_t nbw = (w == pl.W - 1) ? pl.BT : pl.B;
    const uint32_t seg = bx * 256u + threadIdx.x;
    const uint32_t lo = seg * pl.S;
    Xyzz<T> C = xyzz_inf<T>();
    if (lo < nbw) {
        const uint32_t hi = min(lo + pl.S, nbw);
        Xyzz<T> R = xyzz_inf<T>(), Q = xyzz_inf<T>();
        for (uint32_t v = hi; v > lo; v--) {
            const uint32_t g = (uint32_t)w * pl.B + v - 1u;
            const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
            if (t1 > t0) {                                      // multi-task buckets were folded into slot t0
                Xyzz<T> pt = partial[t0];
                pt_add(R, R, pt);
            }
            pt_add(Q, Q, R);
        }
        // sum_{v in (lo, hi]} v * B_v = Q + lo * R
        Xyzz<T> m = xyzz_inf<T>();
        for (int i = 31 - __builtin_clz(lo | 1u); i >= 0; i--) {
            pt_dbl(m, m);
            if ((lo >> i) & 1u) pt_add(m, m, R);
        }
        pt_add(C, Q, m);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off >= 1; off >>= 1) {
        Xyzz<T> o = shfl_from(C, (lane + off) & 63);
        if (lane < off) pt_add(C, C, o);
    }
    __shared__ Xyzz<T> sm[4];
    if (lane == 0) sm[wave] = C;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) pt_add(C, C, sm[k]);
        store_canon<F, T>(&winout[blockIdx.x], C);
    }
}

// ---- two-level bucket reduce (G1, c = 16 plans) -------------------------------------------------------
// The one-lane reduce above is a latency chain: 2 S additions, an offset multiple of ~24 operations and a
// 9-level block tree in a row on 56 K lanes, 0.9 ms for ~9 % of the MSM's field products.  Here the
// weights are split instead: bucket value v = 256 hi + lo + 1, so
//     sum_v v B_v  =  256 sum_hi hi Row_hi  +  sum_lo (lo + 1) Col_lo,      Row_hi = sum_lo B,   Col_lo = sum_hi B
// and the row and column sums are PLAIN sums of the 2^15 buckets of a window (the same two additions per
// bucket as the running sums, but independent): k_msm_rowcol runs them as strided per-lane chains of
// kRcChain buckets plus a short wavefront tree, like the accumulate.  What is left per window are weighted
// sums over 128 + 256 entries: k_msm_reduce_rc, the 4-lane running-sum kernel of the small plans on 32 blocks.
// The host's Horner takes R_w = sum hi Row_hi and C_w = sum (lo + 1) Col_lo as two 8-bit half-windows.
// The unsigned top window (2^16 values) is two half-windows of 2^15 buckets as far as bucket ids go (BT = 2 B), so the
// kernels see W + 1 "virtual windows" of B buckets each; the upper half TB of the top window (virtual window W) only adds
// 128 to its row weights.  16 virtual windows are exactly 1 024 waves of 16-bucket chains -- one per SIMD --, and TB's sums
// (k_msm_rowcol on a second stream) hide behind the accumulate of the other windows: TB's tasks are accumulated first.
static constexpr uint32_t kRcCols = 256;
static constexpr uint32_t kRcRows = 128, kRcPerWindow = kRcRows + kRcCols;                 // B = 32 768 = 128 x 256
// CH: buckets per lane chain -- 16 on a whole MI355X (16 virtual windows = 1 024 waves = one per SIMD), 32 / 64 on a device that reports
// fewer SIMDs (chip_shape), so that the launch on the critical path still places one wave per SIMD
template <uint32_t kRcChain>
__global__ void __launch_bounds__(256)
k_msm_rowcol(const Xyzz<FpL> *__restrict__ bacc, uint32_t B, uint32_t w0, Xyzz<FpL> *__restrict__ rc) {
    // round 4: the buckets' sums are read from the bucket accumulators bacc[bucket] (k_msm_accum_l / k_msm_fold_*) -- no task-offset
    // lookup in front of every point load
    const uint32_t lanes_w = 2u * B / kRcChain;                        // lanes of a virtual window: rows + columns
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t w = w0 + t / lanes_w, local = t % lanes_w;
    const uint32_t row_lanes = kRcRows * (kRcCols / kRcChain);
    const bool is_row = local < row_lanes;                             // uniform in the wave
    const uint32_t lj = is_row ? kRcCols / kRcChain : kRcRows / kRcChain;      // lanes per job: 16 | 8
    const uint32_t l2 = is_row ? local : local - row_lanes, job = l2 / lj, sub = l2 % lj;
    // row job: buckets 256 job + (sub + i lj);   column job: buckets 256 (sub + i lj) + job
    const uint32_t first = w * B + (is_row ? job * kRcCols + sub : sub * kRcCols + job);
    const uint32_t step = is_row ? lj : lj * kRcCols;
    // bucket i + 1's point is loaded during the addition of bucket i: with one wave per SIMD nothing else hides the load
    Xyzz<FpL> acc = bacc[first], nxt = bacc[first + step];
#pragma unroll 1
    for (uint32_t i = 1; i < kRcChain; i++) {
        const Xyzz<FpL> cur = nxt;
        if (i + 1 < kRcChain) nxt = bacc[first + (i + 1u) * step];      // in flight during the addition below
        acc = add(acc, cur);
    }
    const int lane = threadIdx.x & 63;
    for (uint32_t off = lj >> 1; off >= 1; off >>= 1) {                 // the job's lanes are one aligned run of the wave
        const Xyzz<FpL> o = shfl_from(acc, (lane + (int)off) & 63);
        if (sub < off) acc = add(acc, o);
    }
    if (sub == 0) rc[w * kRcPerWindow + (is_row ? job : kRcRows + job)] = acc;
}
// grid = 3 (W + 1) blocks of 64 four-lane groups, each group a run of S = 2 entries:  block 3 w sums (hi + row_weight0) Row_hi over the
// 128 rows, blocks 3 w + 1 / 3 w + 2 sum (lo + 1) Col_lo over the lower / upper 128 columns of virtual window w (round 4: the 256 columns
// were one block with S = 4 -- the longest chain of the kernel; the host adds the two halves).
//   sum_{e in [a, b)} (e + fw + weight0) X_e = Q + (a + weight0) R  with the running sums taken from the top, Q skipping its last addition
// when the weights start at 0.
__global__ void __launch_bounds__(256, 1)
k_msm_reduce_rc(const Xyzz<FpL> *__restrict__ rc, int W, Xyzz<Fp> *__restrict__ winout) {
    const int w = blockIdx.x / 3, kind = blockIdx.x % 3;
    claim_whole_simd();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 3, gb = lane & ~3;
    const uint32_t fw = kind ? 1u : 0u;
    const uint32_t weight0 = kind == 2 ? 128u : (!kind && w == W) ? kRcRows : 0u;          // upper columns | the top window's upper half: rows 128 .. 255
    const Xyzz<FpL> *ent = rc + (size_t)w * kRcPerWindow + (kind ? kRcRows + (kind == 2 ? 128u : 0u) : 0u);
    constexpr uint32_t S = 2u;
    const uint32_t a = (uint32_t)(threadIdx.x >> 2) * S;               // 64 groups x 2 = 128 entries
    Xyzz<FpL> R = xyzz_inf<FpL>(), Q = xyzz_inf<FpL>();
    for (uint32_t e = a + S; e > a; e--) {
        R = add4(R, ent[e - 1], r, gb);
        if (e - 1 > a || fw) Q = add4(Q, R, r, gb);
    }
    Xyzz<FpL> C = add4(Q, small_mul4(R, a + weight0, r, gb), r, gb);
    for (int off = 4; off < 64; off <<= 1) {
        Xyzz<FpL> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 4) C = add4(C, o, r, gb);
    }
    __shared__ Xyzz<FpL> sm[4];
    if (lane == 0) sm[wave] = C;
    __syncthreads();
    if (wave == 0 && lane < 8) {                                       // two groups: (sm[0] + sm[1]) + (sm[2] + sm[3])
        C = add4(sm[2 * (lane >> 2)], sm[2 * (lane >> 2) + 1], r, gb);
        const Xyzz<FpL> o = shfl_from(C, (lane + 4) & 63);
        if (lane < 4) C = add4(C, o, r, gb);
        if (lane == 0) store_canon<Fp, FpL>(&winout[blockIdx.x], C);
    }
}

// ---- two-level bucket reduce of the chain-bound G1 plans (c = 13: 8 193 .. 2^17 records; round 4) -----------------------------
// k_msm_reduce4 is ONE chain per 4-lane group -- 2 S running-sum additions, an offset multiple of ~15 addition-times, a 7-level tree: 0.31 ms of
// a 0.88 ms call at 2^16 records -- and leaves ~230 block sums for the host to add (80 us).  The split of the large plans works here too: a
// window's B = 4 096 buckets are 64 rows x 64 columns (bucket value v = 64 hi + lo + 1; the 9-bit top window: 8 x 64), so
//     sum_v v B_v = 64 sum_hi hi Row_hi + sum_lo (lo + 1) Col_lo.
// k_msm_rowcol_p: plain row / column sums, one lane per chain of kRcpChain buckets (general XYZZ additions in limb form), then a tree over the
// job's 16 (top window's columns: 2) lanes -- 3 + 4 additions deep on 612 waves; k_msm_reduce_rc_p: the weighted sums over 64 (8) entries per
// (window, kind), one entry per 4-lane group.  The host's Horner takes R_w and C_w as a 7-bit and a 6-bit half-window: 40 points instead of 230.
struct RcpGeom { uint32_t B, BT, logC, W; };
static constexpr uint32_t kRcpChain = 4;
__global__ void __launch_bounds__(256)
k_msm_rowcol_p(const Xyzz<Fp> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, RcpGeom g, Xyzz<FpL> *__restrict__ rc) {
    const Xyzz<FpL> *__restrict__ partial = reinterpret_cast<const Xyzz<FpL> *>(partial_);
    const uint32_t C = 1u << g.logC, R = g.B >> g.logC, Rt = g.BT >> g.logC;
    const uint32_t lanes_w = 2u * g.B / kRcpChain, lanes_t = 2u * g.BT / kRcpChain, main_total = (g.W - 1u) * lanes_w;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t w, local, Rw, nb;
    if (t < main_total) { w = t / lanes_w; local = t % lanes_w; Rw = R; nb = g.B; }
    else { local = t - main_total; if (local >= lanes_t) return; w = g.W - 1u; Rw = Rt; nb = g.BT; }      // uniform in the wave (multiples of 64)
    const uint32_t row_lanes = nb / kRcpChain;
    const bool is_row = local < row_lanes;                             // uniform in the wave
    const uint32_t lj = is_row ? C / kRcpChain : Rw / kRcpChain;       // lanes per job
    const uint32_t l2 = is_row ? local : local - row_lanes, job = l2 / lj, sub = l2 % lj;
    // row job: buckets C job + (sub + i lj);   column job: buckets C (sub + i lj) + job
    const uint32_t first = w * g.B + (is_row ? job * C + sub : sub * C + job);
    const uint32_t step = is_row ? lj : lj * C;
    auto fetch = [&](uint32_t i) {
        const uint32_t b = first + i * step, t0 = taskoff[b], t1 = taskoff[b + 1];
        return t1 > t0 ? partial[t0] : xyzz_inf<FpL>();                // multi-task buckets were folded into slot t0
    };
    Xyzz<FpL> acc = fetch(0), nxt = fetch(1);
#pragma unroll 1
    for (uint32_t i = 1; i < kRcpChain; i++) {
        const Xyzz<FpL> cur = nxt;
        if (i + 1 < kRcpChain) nxt = fetch(i + 1);                      // in flight during the addition below
        acc = add(acc, cur);
    }
    const int lane = threadIdx.x & 63;
    for (uint32_t off = lj >> 1; off >= 1; off >>= 1) {                 // the job's lanes are one aligned run of the wave
        const Xyzz<FpL> o = shfl_from(acc, (lane + (int)off) & 63);
        if (sub < off) acc = add(acc, o);
    }
    if (sub == 0) rc[w * (R + C) + (is_row ? job : Rw + job)] = acc;
}
// The same sums with FOUR lanes per addition (add4: the four products of a round dealt over the lanes, ~1/3 of a one-lane addition's latency):
// a 4-lane group per chain of kChain buckets (8: 1 224 waves at two per SIMD; 16: 614 waves), then the tree over the job's groups.  The
// launch is latency-bound (7 one-lane additions in a row on 612 waves), which is what more lanes per addition shorten.
template <uint32_t kChain>
__global__ void __launch_bounds__(256, 2)
k_msm_rowcol_p4(const Xyzz<Fp> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, RcpGeom g, Xyzz<FpL> *__restrict__ rc) {
    const Xyzz<FpL> *__restrict__ partial = reinterpret_cast<const Xyzz<FpL> *>(partial_);
    const uint32_t C = 1u << g.logC, R = g.B >> g.logC, Rt = g.BT >> g.logC;
    const uint32_t col_t = Rt >= kChain ? Rt / kChain : 1u;             // groups of a top-window column job (8 rows: one short chain)
    const uint32_t groups_w = 2u * g.B / kChain, groups_t = Rt * (C / kChain) + C * col_t, main_total = (g.W - 1u) * groups_w;
    const int lane = threadIdx.x & 63, r = lane & 3, gb = lane & ~3;
    const uint32_t t = (blockIdx.x * 256u + threadIdx.x) >> 2;          // group
    uint32_t w, local, Rw, row_groups, lj_col;
    if (t < main_total) { w = t / groups_w; local = t % groups_w; Rw = R; row_groups = g.B / kChain; lj_col = R / kChain; }
    else { local = t - main_total; w = g.W - 1u; Rw = Rt; row_groups = Rt * (C / kChain); lj_col = col_t; }
    const bool live = t < main_total + groups_t;                        // (the grid's last wave: whole groups past the end)
    const bool is_row = local < row_groups;
    const uint32_t lj = is_row ? C / kChain : lj_col;                   // groups per job
    const uint32_t nbj = is_row ? C : Rw, chain = nbj / lj;             // buckets per job, per group
    const uint32_t l2 = is_row ? local : local - row_groups, job = l2 / lj, sub = l2 % lj;
    // row job: buckets C job + (sub + i lj);   column job: buckets C (sub + i lj) + job
    const uint32_t first = w * g.B + (is_row ? job * C + sub : sub * C + job);
    const uint32_t step = is_row ? lj : lj * C;
    auto fetch = [&](uint32_t i) {
        const uint32_t b = first + i * step, t0 = taskoff[b], t1 = taskoff[b + 1];
        return t1 > t0 ? partial[t0] : xyzz_inf<FpL>();                // multi-task buckets were folded into slot t0
    };
    Xyzz<FpL> acc = xyzz_inf<FpL>();
    if (live) {                                                         // uniform in the group
        acc = fetch(0);
        Xyzz<FpL> nxt = chain > 1u ? fetch(1) : xyzz_inf<FpL>();
#pragma unroll 1
        for (uint32_t i = 1; i < chain; i++) {
            const Xyzz<FpL> cur = nxt;
            if (i + 1 < chain) nxt = fetch(i + 1);                      // in flight during the addition below
            acc = add4(acc, cur, r, gb);
        }
    }
    for (uint32_t off = (C / kChain) >> 1; off >= 1; off >>= 1) {       // a job's groups are one aligned run of the wave; column jobs have fewer
        const Xyzz<FpL> o = shfl_from(acc, (lane + 4 * (int)off) & 63);
        if (live && off < lj && sub < off) acc = add4(acc, o, r, gb);
    }
    if (live && sub == 0 && r == 0) rc[w * (R + C) + (is_row ? job : Rw + job)] = acc;
}
// grid = 2 W blocks: block 2 w sums hi Row_hi, block 2 w + 1 sums (lo + 1) Col_lo of window w; 64 four-lane groups, one entry each
__global__ void __launch_bounds__(256, 1)
k_msm_reduce_rc_p(const Xyzz<FpL> *__restrict__ rc, RcpGeom g, Xyzz<Fp> *__restrict__ winout) {
    const uint32_t w = blockIdx.x >> 1, kind = blockIdx.x & 1u;
    claim_whole_simd();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 3, gb = lane & ~3;
    const uint32_t C = 1u << g.logC, R = g.B >> g.logC, Rw = w == g.W - 1u ? g.BT >> g.logC : R;
    const uint32_t n = kind ? C : Rw, e = (uint32_t)(threadIdx.x >> 2);       // entry of this group (n <= 64)
    const Xyzz<FpL> *ent = rc + (size_t)w * (R + C) + (kind ? Rw : 0u);
    Xyzz<FpL> Cs = xyzz_inf<FpL>();
    if (e < n) Cs = small_mul4(ent[e], e + kind, r, gb);               // uniform in the group; weight hi | lo + 1
    for (int off = 4; off < 64; off <<= 1) {
        Xyzz<FpL> o = shfl_from(Cs, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 4) Cs = add4(Cs, o, r, gb);
    }
    __shared__ Xyzz<FpL> sm[4];
    if (lane == 0) sm[wave] = Cs;
    __syncthreads();
    if (wave == 0 && lane < 8) {
        Cs = add4(sm[2 * (lane >> 2)], sm[2 * (lane >> 2) + 1], r, gb);
        const Xyzz<FpL> o = shfl_from(Cs, (lane + 4) & 63);
        if (lane < 4) Cs = add4(Cs, o, r, gb);
        if (lane == 0) store_canon<Fp, FpL>(&winout[blockIdx.x], Cs);
    }
}

// ---- the same two-level split for G2 at c = 13 (BASELINE config 3), on 8-lane groups in limb form (g2_limb.h) ---------------------------
// k_msm_reduce8c_l is one chain of 2 S + ~13 (offset multiple) + 6 (trees) additions of ~12 us per segment, plus a wave per window
// (k_msm_window_sum8c_l) for the block sums.  Rows and columns: k_msm_rowcol8_p gives every job (a row or a column of 64 buckets; the
// top window: 8 rows, columns of 8) four 8-lane groups -- 15 additions and a 2-level tree, 1 228 waves at two per SIMD -- and
// k_msm_reduce_rc8_p one block per (window, kind): group j takes the entries 2 j, 2 j + 1 as  (2 j + kind) (E0 + E1) + E1, then the block tree,
// and leaves the canonical point the host reads (two per window).
struct Rc8Job { uint32_t first, step, chain, lj, sub, out; bool live; };
// groups of a window / of the top window in k_msm_rowcol8_p: rows R (C / kChain) + columns C max(1, R / kChain)
__host__ __device__ inline uint32_t rc8_groups(const RcpGeom &g, uint32_t kChain, bool top) {
    const uint32_t C = 1u << g.logC, R = (top ? g.BT : g.B) >> g.logC;
    return R * (C / kChain) + C * (R >= kChain ? R / kChain : 1u);
}
// group t of k_msm_rowcol8_p -> its chain of buckets and, for the job's first group, the slot of the sum (computed twice -- before the chain
// and again for the store -- so that none of it stays in registers across the additions: the kernel sits at the 256-register line)
template <uint32_t kChain> __device__ __forceinline__ Rc8Job rc8_job(uint32_t t, const RcpGeom &g) {
    const uint32_t C = 1u << g.logC, R = g.B >> g.logC, Rt = g.BT >> g.logC;
    const uint32_t groups_w = rc8_groups(g, kChain, false), groups_t = rc8_groups(g, kChain, true), main_total = (g.W - 1u) * groups_w;
    uint32_t w, local, Rw;
    if (t < main_total) { w = t / groups_w; local = t % groups_w; Rw = R; }
    else { local = t - main_total; w = g.W - 1u; Rw = Rt; }
    const uint32_t row_groups = Rw * (C / kChain), lj_col = Rw >= kChain ? Rw / kChain : 1u;
    Rc8Job j;
    j.live = t < main_total + groups_t;
    const bool is_row = local < row_groups;
    j.lj = is_row ? C / kChain : lj_col;                       // groups per job
    j.chain = (is_row ? C : Rw) / j.lj;                        // buckets per group
    const uint32_t l2 = is_row ? local : local - row_groups, job = l2 / j.lj;
    j.sub = l2 % j.lj;
    // row job: buckets C job + (sub + i lj);   column job: buckets C (sub + i lj) + job
    j.first = w * g.B + (is_row ? job * C + j.sub : j.sub * C + job);
    j.step = is_row ? j.lj : j.lj * C;
    j.out = w * (R + C) + (is_row ? job : Rw + job);
    return j;
}
template <uint32_t kChain>
__global__ void __launch_bounds__(256, 2)
k_msm_rowcol8_p(const Pt2L *__restrict__ partial, const uint32_t *__restrict__ taskoff, RcpGeom g, Pt2L *__restrict__ rc) {
    const int lane = threadIdx.x & 63, sl = lane & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    const uint32_t t = (blockIdx.x * 256u + threadIdx.x) >> 3;          // group
    XyzzK<1> acc = xyzzk_inf<1>();
    {
        const Rc8Job j = rc8_job<kChain>(t, g);
        if (j.live) {                                                   // uniform in the group
#pragma unroll 1
            for (uint32_t i = 0; i < j.chain; i++) {                    // (no prefetch: the next point's 52 registers are the difference between
                const uint32_t b = j.first + i * j.step, t0 = taskoff[b], t1 = taskoff[b + 1];      //  two waves per SIMD and spills)
                if (t1 > t0) acc = add8k(x, acc, load_pt2l(&partial[t0], x.q));                     // multi-task buckets were folded into slot t0
            }
        }
    }
    const uint32_t C = 1u << g.logC;
#pragma unroll 1
    for (uint32_t off = (C / kChain) >> 1; off >= 1; off >>= 1) {       // a job's groups are one aligned run of the wave (column jobs: at most as many)
        const XyzzK<1> o = shfl_from(acc, (lane + 8 * (int)off) & 63);
        const Rc8Job j = rc8_job<kChain>(t, g);
        if (j.live && off < j.lj && j.sub < off) acc = add8k(x, acc, o);
    }
    const Rc8Job j = rc8_job<kChain>(t, g);
    if (j.live && j.sub == 0 && sl < 2) store_pt2l(&rc[j.out], acc, x.q);
}
// grid = 2 W blocks: block 2 w sums hi Row_hi, block 2 w + 1 sums (lo + 1) Col_lo of window w; 32 eight-lane groups, two entries each
__global__ void __launch_bounds__(256)
k_msm_reduce_rc8_p(const Pt2L *__restrict__ rc, RcpGeom g, Xyzz<Fp2> *__restrict__ winout) {
    __shared__ BlockSum8k sm;
    const uint32_t w = blockIdx.x >> 1, kind = blockIdx.x & 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 7;
    const DevLanes8 x{sl >> 1, sl & 1};
    const uint32_t C = 1u << g.logC, R = g.B >> g.logC, Rw = w == g.W - 1u ? g.BT >> g.logC : R;
    const uint32_t n = kind ? C : Rw, e0 = 2u * (uint32_t)(threadIdx.x >> 3);      // entries e0, e0 + 1 of this group (n even, <= 64)
    const Pt2L *ent = rc + (size_t)w * (R + C) + (kind ? Rw : 0u);
    XyzzK<1> Cs = xyzzk_inf<1>();
    if (e0 < n) {                                                       // uniform in the group
        const XyzzK<1> E0 = load_pt2l(&ent[e0], x.q), E1 = load_pt2l(&ent[e0 + 1u], x.q);
        // weights e (rows) | e + 1 (columns):  (e0 + kind) E0 + (e0 + kind + 1) E1 = (e0 + kind) (E0 + E1) + E1
        Cs = add8k(x, small_mul8k(x, add8k(x, E0, E1), e0 + kind), E1);
    }
    Cs = block_sum8k(x, Cs, sm, lane, wave);
    if (threadIdx.x < 2) {
        const Xyzz<FpI> out{to_fpi(Cs.x.l[0]), to_fpi(Cs.y.l[0]), to_fpi(Cs.zz.l[0]), to_fpi(Cs.zzz.l[0])};
        store_component(&winout[blockIdx.x], out, x.q);
    }
}

static void launch_accum(hipStream_t s, uint32_t task_blocks, bool chain_bound, const Aff<Fp> *pts, const void *limb_recs, const uint32_t *entries,
                         const Task *tasks, const uint32_t *perm, const uint32_t *totals, Xyzz<Fp> *partial) {
    const PtL *ptl = reinterpret_cast<const PtL *>(limb_recs);
    if (chain_bound && ptl)
        hipLaunchKernelGGL(k_msm_accum2_l, dim3(task_blocks * 2u), dim3(256), 0, s, ptl, entries, tasks, perm, totals, partial);
    else if (chain_bound)
        hipLaunchKernelGGL(k_msm_accum2<Fp>, dim3(task_blocks * 2u), dim3(256), 0, s, pts, entries, tasks, perm, totals, partial);
    else if (ptl)
        hipLaunchKernelGGL(k_msm_accum_l, dim3(task_blocks), dim3(256), 0, s, ptl, entries, tasks, perm, totals, partial, (const uint32_t *)nullptr);
    else
        hipLaunchKernelGGL(k_msm_accum<Fp>, dim3(task_blocks), dim3(256), 0, s, pts, entries, tasks, perm, totals, partial);
}
static void launch_accum(hipStream_t s, uint32_t task_blocks, bool, const Aff<Fp2> *pts, const void *limb_recs, const uint32_t *entries,
                         const Task *tasks, const uint32_t *perm, const uint32_t *totals, Xyzz<Fp2> *partial) {
    if (limb_recs)
        hipLaunchKernelGGL(k_msm_accum2c_l, dim3(task_blocks * 2u), dim3(256), 0, s, (const PtL2 *)limb_recs, entries, tasks, perm, totals, partial);
    else
        hipLaunchKernelGGL(k_msm_accum2c, dim3(task_blocks * 2u), dim3(256), 0, s, pts, entries, tasks, perm, totals, partial);
}
static void launch_fold_small(hipStream_t s, bool four, bool limb, Xyzz<Fp> *partial, const uint32_t *taskoff, const uint32_t *list, const uint32_t *counts) {
    if (limb && four)
        hipLaunchKernelGGL((k_msm_fold_small4<Fp, FpL>), dim3(2048), dim3(256), 0, s, partial, taskoff, list, counts);
    else if (limb)
        hipLaunchKernelGGL((k_msm_fold_small<Fp, FpL>), dim3(512), dim3(256), 0, s, partial, taskoff, list, counts);
    else if (four)                                  // the plans that take the 4-lane reduce are the latency-bound ones
        hipLaunchKernelGGL(k_msm_fold_small4<Fp>, dim3(2048), dim3(256), 0, s, partial, taskoff, list, counts);
    else
        hipLaunchKernelGGL(k_msm_fold_small<Fp>, dim3(512), dim3(256), 0, s, partial, taskoff, list, counts);
}
static void launch_fold_small(hipStream_t s, bool, bool limb, Xyzz<Fp2> *partial, const uint32_t *taskoff, const uint32_t *list, const uint32_t *counts) {
    if (limb) hipLaunchKernelGGL(k_msm_fold_small8c_l, dim3(1024), dim3(256), 0, s, reinterpret_cast<Pt2L *>(partial), taskoff, list, counts);
    else hipLaunchKernelGGL(k_msm_fold_small8c, dim3(1024), dim3(256), 0, s, partial, taskoff, list, counts);
}
static void launch_fold_big(hipStream_t s, bool limb, Xyzz<Fp> *partial, const uint32_t *taskoff, const uint32_t *list, const uint32_t *counts) {
    if (limb) hipLaunchKernelGGL((k_msm_fold_big<Fp, FpL>), dim3(1024), dim3(256), 0, s, partial, taskoff, list, counts);
    else hipLaunchKernelGGL(k_msm_fold_big<Fp>, dim3(1024), dim3(256), 0, s, partial, taskoff, list, counts);
}
static void launch_fold_big(hipStream_t s, bool limb, Xyzz<Fp2> *partial, const uint32_t *taskoff, const uint32_t *list, const uint32_t *counts) {
    if (limb) hipLaunchKernelGGL(k_msm_fold_big8c_l, dim3(1024), dim3(256), 0, s, reinterpret_cast<Pt2L *>(partial), taskoff, list, counts);
    else hipLaunchKernelGGL(k_msm_fold_big<Fp2>, dim3(1024), dim3(256), 0, s, partial, taskoff, list, counts);
}
static void launch_reduce(hipStream_t s, uint32_t red_blocks, bool four, bool limb, const Xyzz<Fp> *partial, const uint32_t *taskoff,
                          const MsmPlan &pl, const ReduceGrid &rg, Xyzz<Fp> *winout, const RcpGeom *rcp, Xyzz<FpL> *rc) {
    if (rcp) {                  // c = 13: row / column sums, then the weighted sums (two launches; red_blocks = 2 W)
        // EIP2537_RCP_CHAIN=4: one lane per chain of 4 buckets (k_msm_rowcol_p); 8 | 16: a 4-lane group per chain (k_msm_rowcol_p4)
        static const uint32_t env_chain = [] { const char *v = getenv("EIP2537_RCP_CHAIN"); return v ? (uint32_t)atoi(v) : 0u; }();
        const uint32_t C = 1u << rcp->logC, Rt = rcp->BT >> rcp->logC;
        const bool small = C < 64u;                         // the c = 8 plans (8 x 16 buckets per window): 4-lane groups per chain of 4
        const uint32_t chain = small ? 0u : env_chain == 4u || env_chain == 8u || env_chain == 16u ? env_chain : 8u;
        if (small) {
            const uint32_t groups = (rcp->W - 1u) * (2u * rcp->B / 4u) + Rt * (C / 4u) + C * (Rt / 4u);
            hipLaunchKernelGGL(k_msm_rowcol_p4<4u>, dim3((groups * 4u + 255u) / 256u), dim3(256), 0, s, partial, taskoff, *rcp, rc);
        } else if (chain == 4u) {
            const uint32_t lanes = (rcp->W - 1u) * (2u * rcp->B / kRcpChain) + 2u * rcp->BT / kRcpChain;
            hipLaunchKernelGGL(k_msm_rowcol_p, dim3((lanes + 255u) / 256u), dim3(256), 0, s, partial, taskoff, *rcp, rc);
        } else {
            const uint32_t groups = (rcp->W - 1u) * (2u * rcp->B / chain) + Rt * (C / chain) + C * (Rt >= chain ? Rt / chain : 1u);
            const dim3 grid((groups * 4u + 255u) / 256u);
            if (chain == 8u) hipLaunchKernelGGL(k_msm_rowcol_p4<8u>, grid, dim3(256), 0, s, partial, taskoff, *rcp, rc);
            else hipLaunchKernelGGL(k_msm_rowcol_p4<16u>, grid, dim3(256), 0, s, partial, taskoff, *rcp, rc);
        }
        hipLaunchKernelGGL(k_msm_reduce_rc_p, dim3(red_blocks), dim3(256), 0, s, (const Xyzz<FpL> *)rc, *rcp, winout);
        return;
    }
    if (limb && four) hipLaunchKernelGGL((k_msm_reduce4<Fp, FpL>), dim3(red_blocks), dim3(256), 0, s, partial, taskoff, pl, rg, winout);
    else if (limb) hipLaunchKernelGGL((k_msm_reduce1<Fp, FpL>), dim3(red_blocks), dim3(256), 0, s, partial, taskoff, pl, rg, winout);
    else if (four) hipLaunchKernelGGL(k_msm_reduce4<Fp>, dim3(red_blocks), dim3(256), 0, s, partial, taskoff, pl, rg, winout);
    else hipLaunchKernelGGL(k_msm_reduce1<Fp>, dim3(red_blocks), dim3(256), 0, s, partial, taskoff, pl, rg, winout);
}
static void launch_reduce(hipStream_t s, uint32_t red_blocks, bool, bool limb, const Xyzz<Fp2> *partial, const uint32_t *taskoff,
                          const MsmPlan &pl, const ReduceGrid &rg, Xyzz<Fp2> *winout, const RcpGeom *rcp, Xyzz<FpL> *rc) {
    if (rcp) {                  // c = 13, limb form: row / column sums, then the weighted sums (red_blocks = 2 W canonical points)
        Pt2L *rc8 = reinterpret_cast<Pt2L *>(rc);
        if ((1u << rcp->logC) < 64u) {      // the c = 8 plans (8 x 16 buckets per window): chains of 4
            const uint32_t groups = (rcp->W - 1u) * rc8_groups(*rcp, 4u, false) + rc8_groups(*rcp, 4u, true);
            hipLaunchKernelGGL(k_msm_rowcol8_p<4u>, dim3((groups * 8u + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const Pt2L *>(partial), taskoff, *rcp, rc8);
        } else {
            const uint32_t groups = (rcp->W - 1u) * rc8_groups(*rcp, 16u, false) + rc8_groups(*rcp, 16u, true);
            hipLaunchKernelGGL(k_msm_rowcol8_p<16u>, dim3((groups * 8u + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<const Pt2L *>(partial), taskoff, *rcp, rc8);
        }
        hipLaunchKernelGGL(k_msm_reduce_rc8_p, dim3(red_blocks), dim3(256), 0, s, (const Pt2L *)rc8, *rcp, winout);
        return;
    }
    if (limb) {                 // limb-form block sums live behind the W canonical window sums
        Pt2L *blkout = reinterpret_cast<Pt2L *>(winout + red_blocks + pl.W);
        hipLaunchKernelGGL(k_msm_reduce8c_l, dim3(red_blocks), dim3(256), 0, s, reinterpret_cast<const Pt2L *>(partial), taskoff, pl, rg, blkout);
        hipLaunchKernelGGL(k_msm_window_sum8c_l, dim3(pl.W), dim3(64), 0, s, (const Pt2L *)blkout, pl, rg, winout + red_blocks);
        return;
    }
    hipLaunchKernelGGL(k_msm_reduce8c, dim3(red_blocks), dim3(256), 0, s, partial, taskoff, pl, rg, winout);
    hipLaunchKernelGGL(k_msm_window_sum8c, dim3(pl.W), dim3(64), 0, s, (const Xyzz<Fp2> *)winout, pl, rg, winout + red_blocks);     // -> W window sums behind the block sums
}
static constexpr bool window_sums_on_device(const Fp2 *) { return true; }
// (round 4, measured: the same for the G1 c <= 13 plans -- block sums kept in limb form, one more wave per window, k_msm_window_sum4 --
// is neutral: the kernel costs the device the 40-80 us it saves the host (2^16: 0.887 -> 0.872 ms, 2^12: 0.631 -> 0.639, 128 records
// 0.423 -> 0.421; profiles/r04_size_sweep.txt).  Not kept.)
static constexpr bool window_sums_on_device(const Fp *) { return false; }
// Two-level reduce of a G1 c = 16 plan (k_msm_rowcol / k_msm_reduce_rc), behind the accumulate of the call's LAST record shard: the top
// window's upper half TB is accumulated, folded and summed on stream3 beside the accumulate of everything else (TB's tasks finish long
// before the rest: 1 / 17 of the work); the row / column launch on the critical path then is the 16 other virtual windows = exactly
// 1 024 waves.  Every bucket's sum over all shards of the call lives in bacc (k_msm_accum_l).
struct TwoLevelArgs {
    const PtL *ptl; const uint32_t *entries; const Task *tasks; const uint32_t *perm, *totals, *ranges; Xyzz<Fp> *partial; const uint32_t *taskoff;
    const uint32_t *split_small, *split_big, *taskbkt; Xyzz<FpL> *bacc, *rc; Xyzz<Fp> *winout; uint32_t first_shard;
};
static int launch_two_level(Engine *e, const MsmPlan &pl, size_t ns, uint32_t lshift, uint32_t task_blocks, uint32_t red_blocks, uint32_t split_g,
                            const TwoLevelArgs &a) {
    hipStream_t s = e->stream;
    const uint32_t tb_blocks = (pl.B + (uint32_t)(ns >> lshift) + 1u + 255u) / 256u;     // at most B buckets + ns / L full tasks
    // chain length: the shortest whose W virtual windows place at most one wave per SIMD (16 on 1 024 SIMDs)
    const uint32_t simds = chip_shape(e->device).simds;
    const uint32_t chain = (uint32_t)pl.W * (2u * pl.B / 16u / 64u) <= simds ? 16u : (uint32_t)pl.W * (2u * pl.B / 32u / 64u) <= simds ? 32u : 64u;
    const uint32_t unit_blocks = (2u * pl.B / chain) / 256u;                             // blocks of one virtual window in k_msm_rowcol
    auto rowcol = [&](hipStream_t st, uint32_t blocks, uint32_t w0) {
        if (chain == 16u) hipLaunchKernelGGL(k_msm_rowcol<16u>, dim3(blocks), dim3(256), 0, st, (const Xyzz<FpL> *)a.bacc, pl.B, w0, a.rc);
        else if (chain == 32u) hipLaunchKernelGGL(k_msm_rowcol<32u>, dim3(blocks), dim3(256), 0, st, (const Xyzz<FpL> *)a.bacc, pl.B, w0, a.rc);
        else hipLaunchKernelGGL(k_msm_rowcol<64u>, dim3(blocks), dim3(256), 0, st, (const Xyzz<FpL> *)a.bacc, pl.B, w0, a.rc);
    };
    // stream3, beside the main accumulate (a launch of TB's ~770 waves alone would be a latency chain on an empty chip: 0.25 ms lost)
    HIPCHK(e->need_stream3());
    hipStream_t s3 = e->stream3;
    HIPCHK(hipEventRecord(e->ev_j3, s));
    HIPCHK(hipStreamWaitEvent(s3, e->ev_j3, 0));
    hipLaunchKernelGGL(k_msm_accum_l, dim3(tb_blocks), dim3(256), 0, s3, a.ptl, a.entries, a.tasks, a.perm, a.totals, a.partial, a.ranges + 2, a.taskbkt, a.bacc, a.first_shard);
    hipLaunchKernelGGL((k_msm_fold_small<Fp, FpL>), dim3(512), dim3(256), 0, s3, a.partial, a.taskoff, a.split_small, a.totals + 2, split_g, 0xffffffffu, a.bacc);
    hipLaunchKernelGGL((k_msm_fold_big<Fp, FpL>), dim3(1024), dim3(256), 0, s3, a.partial, a.taskoff, a.split_big, a.totals + 2, split_g, 0xffffffffu, a.bacc);
    // the side chain has the chip's other SIMDs to spare: half-length chains on twice the waves (128 instead of 64) shorten it by ~0.12 ms,
    // which matters behind the SHORT last accumulate of a staged call (its row / column sums ran 60 us into the main launch and slowed it)
    if (chain == 16u) hipLaunchKernelGGL(k_msm_rowcol<8u>, dim3(2u * unit_blocks), dim3(256), 0, s3, (const Xyzz<FpL> *)a.bacc, pl.B, (uint32_t)pl.W, a.rc);
    else rowcol(s3, unit_blocks, (uint32_t)pl.W);
    HIPCHK(hipEventRecord(e->ev_j2, s3));
    hipLaunchKernelGGL(k_msm_accum_l, dim3(task_blocks), dim3(256), 0, s, a.ptl, a.entries, a.tasks, a.perm, a.totals, a.partial, a.ranges, a.taskbkt, a.bacc, a.first_shard);
    HIPCHK(hipEventRecord(e->ev_b, s));
    hipLaunchKernelGGL((k_msm_fold_small<Fp, FpL>), dim3(512), dim3(256), 0, s, a.partial, a.taskoff, a.split_small, a.totals + 2, 0u, split_g, a.bacc);
    hipLaunchKernelGGL((k_msm_fold_big<Fp, FpL>), dim3(1024), dim3(256), 0, s, a.partial, a.taskoff, a.split_big, a.totals + 2, 0u, split_g, a.bacc);
    rowcol(s, unit_blocks * (uint32_t)pl.W, 0u);                                         // 16 units: 1 024 waves
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    hipLaunchKernelGGL(k_msm_reduce_rc, dim3(red_blocks), dim3(256), 0, s, (const Xyzz<FpL> *)a.rc, pl.W, a.winout);
    return E_SUCCESS;
}
// an earlier shard of a staged call: all its tasks in one launch, onto the bucket accumulators
static void launch_shard_accum(hipStream_t s, uint32_t task_blocks, const TwoLevelArgs &a) {
    hipLaunchKernelGGL(k_msm_accum_l, dim3(task_blocks), dim3(256), 0, s, a.ptl, a.entries, a.tasks, a.perm, a.totals, a.partial, (const uint32_t *)nullptr, a.taskbkt, a.bacc, a.first_shard);
    hipLaunchKernelGGL((k_msm_fold_small<Fp, FpL>), dim3(512), dim3(256), 0, s, a.partial, a.taskoff, a.split_small, a.totals + 2, 0u, 0xffffffffu, a.bacc);
    hipLaunchKernelGGL((k_msm_fold_big<Fp, FpL>), dim3(1024), dim3(256), 0, s, a.partial, a.taskoff, a.split_big, a.totals + 2, 0u, 0xffffffffu, a.bacc);
}
static int launch_two_level_or_shard(Engine *e, const MsmPlan &ps, size_t ns, uint32_t lshift, uint32_t task_blocks, uint32_t red_blocks, uint32_t split_g, bool last,
                                     const PtL *ptl, const uint32_t *entries, const Task *tasks, const uint32_t *perm, const uint32_t *totals, const uint32_t *ranges,
                                     Xyzz<Fp> *partial, const uint32_t *taskoff, const uint32_t *split_small, const uint32_t *split_big, const uint32_t *taskbkt,
                                     Xyzz<FpL> *bacc, Xyzz<FpL> *rc, Xyzz<Fp> *winout, uint32_t first_shard) {
    const TwoLevelArgs a{ptl, entries, tasks, perm, totals, ranges, partial, taskoff, split_small, split_big, taskbkt, bacc, rc, winout, first_shard};
    if (last) return launch_two_level(e, ps, ns, lshift, task_blocks, red_blocks, split_g, a);
    launch_shard_accum(e->stream, task_blocks, a);
    return E_SUCCESS;
}
static int launch_two_level_or_shard(Engine *, const MsmPlan &, size_t, uint32_t, uint32_t, uint32_t, uint32_t, bool, const PtL *, const uint32_t *, const Task *,
                                     const uint32_t *, const uint32_t *, const uint32_t *, Xyzz<Fp2> *, const uint32_t *, const uint32_t *, const uint32_t *,
                                     const uint32_t *, Xyzz<FpL> *, Xyzz<FpL> *, Xyzz<Fp2> *, uint32_t) { return E_MEMORY_ERROR; }      // (never selected for G2)
static constexpr uint32_t kFourLaneMaxBuckets = 131072;
static constexpr uint32_t kMinTaskShift = 4;        // c <= 13: tasks of at most max(16, 2 x mean bucket load) entries
template <class F> struct ReduceCfg { static constexpr bool kFourLane = false; static constexpr const char *kName = "eip::Fp"; };
template <> struct ReduceCfg<Fp2> { static constexpr bool kFourLane = true; static constexpr const char *kName = "eip::Fp2"; };

// One device pipeline.  Host input (e->host_src): the pipeline stages the caller's records itself -- ONE copy in front of the
// kernels, or, for the large G1 plans when api.hip has cut the call into record shards (e->feed, round 4), shard by shard: the slot's
// helper thread hands the shards to the copy stream back to back (StagedCopy, engine.h) while this thread launches, behind each
// shard's copy event, that shard's decode, sort and accumulate; the shards share ONE bucket space (bacc: every bucket's running sum
// over the shards), so the bucket reduce and the host tail run once, behind the last shard.  The 168 MB copy of 2^20 records
// (3.2 ms at the link's ~53 GB/s) is as long as the whole device pipeline; round 3 ran the shards as independent pipelines
// (k sorts, k reduces over the same 557 056 buckets, k host tails: profiles/r03_h2d_pipeline.txt).
template <class F>
static int msm_device_t(Engine *e, const void *d_in, size_t n, uint32_t *partial_words, int c_override, bool direct_sort = false) {
    ShardFeed feed = e->feed;                   // (a staged call's cut; consumed here whatever happens below)
    e->feed.k = 0;
    const void *host_src = e->host_src;
    e->host_src = nullptr;                      // never retained past the call
    StagedCopy staged_copy;                     // before anything that can return: its destructor waits for the helper thread
    if (n == 0 || n >= (1ull << 31)) return E_MEMORY_ERROR;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) != 0) {
        fprintf(stderr, "[eip2537_hip] device input must be 4-byte aligned\n");
        return E_MEMORY_ERROR;
    }
    MsmPlan pl = msm_make_plan((uint32_t)n, c_override, ReduceCfg<F>::kFourLane);
    if (pl.max_entries >= (1ull << 32)) return E_MEMORY_ERROR;
    // reduce: a latency-bound serial chain of ~2S + 30 point operations per segment.
    //  G1: one lane per segment, 256 segments per block
    //  G2: 4 lanes per segment, 64 segments per block; the shortest chain that still places at
    //      most one wave on every SIMD (<= ~232 working blocks on 256 CUs: a second wave on a SIMD
    //      doubles the latency of both and the kernel waits for the slowest)
    // G1 takes the 4-lane form too while the bucket count is small (c <= 13 plans): few waves, so the
    // shorter chain wins (2^7 .. 2^16 records: 0.61 -> 0.41 ms .. 1.76 -> 1.45 ms of device time); at
    // c = 16 (557 K buckets) four lanes per segment would oversubscribe the SIMDs and lose.
    const bool four = ReduceCfg<F>::kFourLane || pl.NB <= kFourLaneMaxBuckets;
    const bool eight = ReduceCfg<F>::kFourLane;               // G2: 8 lanes per segment, split by component
    const uint32_t seg_per_block = eight ? 32u : four ? 64u : 256u;
    // Segment length S: the shortest chain whose blocks (4 waves each, one wave per SIMD: both kernels
    // claim the whole SIMD) still fit the chip in one round -- 256 CUs x 4 SIMDs.  The grid holds
    // working blocks only, so the target is close to that limit (before: a (blocks, W) grid sized for
    // the top window, half of it idle, and ~140 working blocks: 2^20 reduce 1.42 ms, S = 16).
    static const uint32_t env_rb = [] { const char *v = getenv("EIP2537_REDUCE_BLOCKS"); return v ? (uint32_t)atoi(v) : 0u; }();
    const ChipShape chip = chip_shape(e->device);
    const uint32_t block_target = env_rb ? env_rb : (four ? chip.cus - chip.cus * 3u / 32u : chip.cus - chip.cus / 42u);   // 232 | 250 of 256 CUs
    pl.S = std::max(four ? 1u : 2u, (pl.NB + seg_per_block * block_target - 1u) / (seg_per_block * block_target));
    ReduceGrid rg;
    uint32_t red_blocks;
    for (;; pl.S++) {                          // per-window rounding can push the grid past one wave per SIMD: lengthen S then
        rg.bn = ((pl.B + pl.S - 1) / pl.S + seg_per_block - 1u) / seg_per_block;
        rg.bt = ((pl.BT + pl.S - 1) / pl.S + seg_per_block - 1u) / seg_per_block;
        red_blocks = (uint32_t)(pl.W - 1) * rg.bn + rg.bt;
        if (red_blocks <= chip.cus || env_rb || pl.S >= 4096u) break;
    }
    // G1 plans run accumulate, fold and reduce in limb form (limb30.h): the decode kernel writes 168-byte
    // limb records instead of the 96-byte affine points.  EIP2537_LIMB_FORM=0: the FpI kernels.
    static const bool env_limb = [] { const char *v = getenv("EIP2537_LIMB_FORM"); return !v || atoi(v) != 0; }();
    const bool limb_form = !ReduceCfg<F>::kFourLane && env_limb && (pl.c > 13 ? !four : four);     // G1: (one lane, reduce1) or (two lanes, reduce4)
    // G1, c = 16: the two-level reduce (row / column sums k_msm_rowcol, then 2 (W + 1) small weighted sums k_msm_reduce_rc);
    // EIP2537_REDUCE_RC=0: the one-lane chain k_msm_reduce1.  Round 3, first form (profiles/r03_two_level_reduce.txt): all 17
    // half-window units in one launch are 1 088 waves for 1 024 SIMDs -- 0.60 + 0.24 ms, no gain over the chain's 0.89 ms.  Final
    // form: the upper half of the top window is accumulated FIRST and its sums run on a second stream behind the accumulate of the
    // rest, so that the launch on the critical path is exactly 1 024 waves (0.37 ms).
    static const bool env_rc = [] { const char *v = getenv("EIP2537_REDUCE_RC"); return !v || atoi(v) != 0; }();
    const bool two_level = limb_form && env_rc && !four && pl.c == 16 && pl.B == kRcRows * kRcCols && pl.BT == 2u * pl.B;
    if (two_level) red_blocks = 3u * (uint32_t)(pl.W + 1);
    // G1, c = 13 (8 193 .. 2^17 records): the same split for the chain-bound plans (k_msm_rowcol_p / k_msm_reduce_rc_p); EIP2537_REDUCE_RCP=0:
    // the 4-lane running-sum chain k_msm_reduce4
    static const bool env_rcp = [] { const char *v = getenv("EIP2537_REDUCE_RCP"); return !v || atoi(v) != 0; }();
    // geometries: c = 13 -- 4 096 = 64 x 64 buckets per window, top window 8 x 64; c = 8 (up to 2 048 records) -- 128 = 8 x 16, top window 16 x 16
    const bool rcp_geom13 = pl.c == 13 && pl.B == 4096u && pl.BT == 512u, rcp_geom8 = pl.c == 8 && pl.B == 128u && pl.BT == 256u;
    const RcpGeom rcp{pl.B, pl.BT, rcp_geom8 ? 4u : 6u, (uint32_t)pl.W};
    // G2: accumulate, fold and reduce in limb form too (k_msm_accum2c_l, g2_limb.h); EIP2537_G2_LIMB=0: the FpI kernels
    static const bool env_g2limb = [] { const char *v = getenv("EIP2537_G2_LIMB"); return !v || atoi(v) != 0; }();
    const bool g2_limb = ReduceCfg<F>::kFourLane && env_g2limb;
    // G2, c = 13: k_msm_rowcol8_p / k_msm_reduce_rc8_p; EIP2537_REDUCE_RCP8=0: the running-sum chain k_msm_reduce8c_l
    static const bool env_rcp8 = [] { const char *v = getenv("EIP2537_REDUCE_RCP8"); return !v || atoi(v) != 0; }();
    // (G2 at c = 8 lost 30 us per call while the host added its twice-as-many window sums in scalar code; with the Horner accumulator kept
    //  in one IFMA vector, additions included, it gains: profiles/r04_reduce_rcp.txt)
    const bool two_level_p = !two_level && (rcp_geom13 || rcp_geom8) &&
                             (ReduceCfg<F>::kFourLane ? (g2_limb && env_rcp8) : (limb_form && four && env_rcp));
    if (two_level_p) red_blocks = 2u * (uint32_t)pl.W;
    const bool dev_winsum = window_sums_on_device((const F *)nullptr) && !two_level && !two_level_p;      // G2: one sum per window comes back, not one per block
    const size_t nwin_out = dev_winsum ? (size_t)pl.W : red_blocks;
    const size_t rc_bytes = two_level ? (size_t)(pl.W + 1) * kRcPerWindow * sizeof(Xyzz<FpL>) : 0;
    const uint32_t split_top = two_level ? (uint32_t)pl.W * pl.B : 0xffffffffu;      // first bucket of the top window's upper half

    // Record shards: [bound[s], bound[s + 1]).  Only a two-level plan can share its buckets between shards; anything else takes
    // the caller's buffer in one copy.
    const bool sharded = two_level && feed.k > 1 && feed.k <= ShardFeed::kMax && feed.bound[0] == 0u && feed.bound[feed.k] == (uint32_t)n;
    const bool staged = sharded && host_src;           // the shards are copied from the caller's buffer one behind the other
    if (!sharded) { feed.k = 1; feed.bound[0] = 0u; feed.bound[1] = (uint32_t)n; }
    const int K = feed.k;
    // task length limit L = 2^lshift: at least twice the mean bucket load so that split buckets stay
    // the exception (they cost an extra fold pass), and at least kMinTaskShift.  Below ~2^18 records
    // the accumulate is not throughput-bound: its time is the longest task's chain (~12 us per mixed
    // addition), set by the few heavy buckets of the short top window (c = 13: 512 buckets holding
    // n/512 records each), so a shorter L there trades a few fold additions for a 2-4x shorter chain.
    // (measured, min 64 -> 16: G1 2^14 1.20 -> 0.86 ms, 2^16 1.44 -> 1.15 ms, G2 2^16 3.25 -> 2.80 ms of
    // device time; L below twice the mean load is far worse -- the fold pass then sees most buckets)
    // (only for the c <= 13 plans, whose short top window is the chain in question; at c = 16 the
    // floor of 64 changes nothing for ordinary inputs and keeps one-bucket adversarial inputs at
    // 2.7 instead of 3.8 ms for 2^18 records)
    auto shift_for = [&](uint32_t ns) {
        uint32_t l = pl.c <= 13 ? kMinTaskShift : 6u;
        while (l < 20 && (1ull << l) < 2ull * ns / pl.B) l++;
        return l;
    };
    uint32_t ns_max = 0, max_tasks = 0;
    for (int sh = 0; sh < K; sh++) {
        if (feed.bound[sh + 1] <= feed.bound[sh]) return E_MEMORY_ERROR;
        const uint32_t ns = feed.bound[sh + 1] - feed.bound[sh];
        ns_max = std::max(ns_max, ns);
        max_tasks = std::max(max_tasks, (uint32_t)(pl.NB + (((uint64_t)ns * pl.W) >> shift_for(ns)) + 1u));
    }
    pl.L = 1u << shift_for(ns_max);
    pl.max_tasks = max_tasks;
    // Sharded call: the SORT stage of shard s + 1 (decode .. task order: latency- and LDS-bound kernels that leave the vector units idle)
    // runs on stream3 beside the ACCUMULATE of shard s on the main stream.  Everything the accumulate / fold kernels read from the sort
    // stage exists twice (by shard parity); events order sort(s) -> accumulate(s) and accumulate(s) -> sort(s + 2).
    // EIP2537_SORT_OVERLAP=0: one stream, one set (A/B).
    static const bool env_overlap = [] { const char *v = getenv("EIP2537_SORT_OVERLAP"); return !v || atoi(v) != 0; }();
    const bool overlap = sharded && !host_src && K > 1 && env_overlap;
    const size_t dup = overlap ? 2 : 1;
    const size_t pt_bytes = limb_form ? sizeof(PtL) : g2_limb ? sizeof(PtL2) : sizeof(Aff<F>);
    HIPCHK(e->pts.reserve(dup * (size_t)ns_max * pt_bytes));
    HIPCHK(e->counts.reserve((size_t)pl.NB * 4));
    HIPCHK(e->offsets.reserve((size_t)pl.NB * 4));
    const uint32_t nbmax = (std::max(pl.B, pl.BT) + 1u) & ~1u;
    HIPCHK(e->digits.reserve((size_t)pl.W * ns_max * 4));                          // digits [W][ns]
    // c = 16 plans below 2^24 records: the partitioned sort (k_sort_*); EIP2537_SORT2=0 or any other plan: direct scatter
    static const bool env_sort2 = [] { const char *v = getenv("EIP2537_SORT2"); return !v || atoi(v) != 0; }();
    const bool sort2 = env_sort2 && !direct_sort && pl.c == 16 && ns_max < (1u << 24) && (pl.B % kFine) == 0 && (pl.BT % kFine) == 0 &&
                       (std::max(pl.B, pl.BT) >> kFineBits) <= kMaxParts && pl.W < 64;
    // hist16: [W][slices][nbmax] packed slice histograms (direct scatter) | [W][parts][slices] partition counts (partitioned sort);
    // slice_base: [W][slices][nbmax] prefix over the slices | the entries in partition order.  Slices: msm_slice_for, per shard.
    size_t hist_bytes = 0, base_bytes = 0;
    for (int sh = 0; sh < K; sh++) {
        const uint32_t ns = feed.bound[sh + 1] - feed.bound[sh], sl = msm_slice_for(ns, pl.W, sort2), nsl = (ns + sl - 1u) / sl;
        hist_bytes = std::max(hist_bytes, sort2 ? (size_t)pl.W * kMaxParts * nsl * 4 : (size_t)pl.W * nsl * (nbmax / 2) * 4);
        base_bytes = std::max(base_bytes, sort2 ? (size_t)pl.W * ns * 4 : (size_t)pl.W * nsl * nbmax * 4);
    }
    HIPCHK(e->hist16.reserve(hist_bytes));
    HIPCHK(e->slice_base.reserve(base_bytes));
    HIPCHK(e->taskoff.reserve(dup * (size_t)(pl.NB + 1) * 4));
    HIPCHK(e->entries.reserve(dup * (size_t)pl.W * ns_max * 4));
    HIPCHK(e->tasks.reserve(dup * (size_t)pl.max_tasks * sizeof(Task)));
    HIPCHK(e->partial.reserve((size_t)pl.max_tasks * (limb_form ? sizeof(Xyzz<FpL>) : g2_limb ? sizeof(Pt2L) : sizeof(Xyzz<F>))));
    HIPCHK(e->winout.reserve(((size_t)red_blocks + (size_t)pl.W) * sizeof(Xyzz<F>) + (g2_limb ? (size_t)red_blocks * sizeof(Pt2L) : 0)));
    HIPCHK(e->misc.reserve(64));                            // first-error word | heavy flag | totals (x2)
    HIPCHK(e->scan_blk.reserve(dup * kScanBlkWords * 4));   // scan block totals, task-length histograms / offsets (two sets), slot ranges, window totals / bases
    HIPCHK(e->perm.reserve(dup * (size_t)pl.max_tasks * 4));
    HIPCHK(e->split_lists.reserve(dup * (size_t)pl.NB * 8));      // split-bucket lists: small | big
    if (two_level_p) {          // per window R row + C column sums (the top window may have more rows than the others)
        const size_t per_w = (size_t)(std::max(rcp.B, rcp.BT) >> rcp.logC) + ((size_t)1 << rcp.logC);
        HIPCHK(e->rcsum.reserve((size_t)pl.W * per_w * (g2_limb ? sizeof(Pt2L) : sizeof(Xyzz<FpL>))));
    }
    if (two_level) {
        HIPCHK(e->bacc.reserve((size_t)pl.NB * sizeof(Xyzz<FpL>)));
        HIPCHK(e->taskbkt.reserve(dup * (size_t)pl.max_tasks * 4));
        HIPCHK(e->rcsum.reserve(rc_bytes));
    }
    if ((pl.NB + 1023u) / 1024u > 1024u) return E_MEMORY_ERROR;

    static const uint32_t env_sp = [] { const char *v = getenv("EIP2537_SCATTER_PASSES"); return v ? (uint32_t)atoi(v) : 0u; }();
    hipStream_t s = e->stream;
    auto *err = reinterpret_cast<unsigned long long *>(e->misc.p);
    uint32_t *heavy = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 8);        // raised by k_sort_coarse_scan: degenerate input
    hipStream_t ss = s;                          // the stream of the sort stage
    if (overlap) {
        HIPCHK(e->need_stream3());
        ss = e->stream3;
        for (int sh = 0; sh < K; sh++) {
            if (!e->ev_sorted[sh]) HIPCHK(hipEventCreateWithFlags(&e->ev_sorted[sh], hipEventDisableTiming));
            if (!e->ev_accdone[sh]) HIPCHK(hipEventCreateWithFlags(&e->ev_accdone[sh], hipEventDisableTiming));
        }
    }
    HIPCHK(hipMemsetAsync(e->misc.p, 0xFF, 8, ss));      // (on the stream whose first kernel, the decode, writes the error word)
    HIPCHK(hipMemsetAsync(heavy, 0, 8, ss));

    const size_t rec_words = Wire<F>::kMsmRecWords, rec_bytes = rec_words * 4;
    auto *digits = reinterpret_cast<uint32_t *>(e->digits.p);
    auto *counts = reinterpret_cast<uint32_t *>(e->counts.p);
    auto *offsets = reinterpret_cast<uint32_t *>(e->offsets.p);
    auto *hist16 = reinterpret_cast<uint32_t *>(e->hist16.p);
    auto *base = reinterpret_cast<uint32_t *>(e->slice_base.p);
    auto *partial = reinterpret_cast<Xyzz<F> *>(e->partial.p);
    auto *winout = reinterpret_cast<Xyzz<F> *>(e->winout.p);
    Xyzz<FpL> *bacc = two_level ? reinterpret_cast<Xyzz<FpL> *>(e->bacc.p) : nullptr;

    {
        const bool two_lane = ReduceCfg<F>::kFourLane || pl.c <= 13;
        LastPlan lp{};
        if (ReduceCfg<F>::kFourLane) snprintf(lp.kernel, sizeof lp.kernel, g2_limb ? "k_msm_accum2c_l" : "k_msm_accum2c");        // G2: split by component
        else if (limb_form) snprintf(lp.kernel, sizeof lp.kernel, two_lane ? "k_msm_accum2_l" : "k_msm_accum_l");
        else snprintf(lp.kernel, sizeof lp.kernel, "%s<%s>", two_lane ? "k_msm_accum2" : "k_msm_accum", ReduceCfg<F>::kName);
        lp.c = pl.c; lp.windows = pl.W; lp.lanes = two_lane ? 2 : 1; lp.units = (uint32_t)n; lp.buckets = pl.NB; lp.shards = K;
        e->last_plan = lp;
    }
    bool inline_copies = false;
    if (staged) {
        HIPCHK(e->need_stream2());
        for (int sh = 0; sh < K; sh++)
            if (!e->ev_copy[sh]) HIPCHK(hipEventCreateWithFlags(&e->ev_copy[sh], hipEventDisableTiming));
        // the helper thread copies; without one (no thread to be had) this thread copies each shard itself in front of its kernels
        inline_copies = !staged_copy.start(e->helper, e->device, e->stream2, e->ev_copy, feed, rec_bytes, const_cast<void *>(d_in), host_src);
    }
    HIPCHK(hipEventRecord(e->ev_start, s));
    for (int sh = 0; sh < K; sh++) {
        const uint32_t r0 = feed.bound[sh], ns = feed.bound[sh + 1] - r0;
        const bool last = sh == K - 1;
        const uint32_t first_shard = sh == 0 ? 1u : 0u;
        MsmPlan ps = pl;                                   // the shard's own plan: same windows and buckets, its own records
        ps.n = ns;
        const uint32_t lshift = shift_for(ns);
        const uint32_t gshift = lshift > 6 ? lshift - 6 : 0;   // granularity of the 64 task-length classes
        ps.L = 1u << lshift;
        ps.max_entries = (uint64_t)ns * pl.W;
        ps.max_tasks = (uint32_t)(pl.NB + (ps.max_entries >> lshift) + 1u);
        ps.slice = msm_slice_for(ns, pl.W, sort2);
        const uint32_t nslices = (ns + ps.slice - 1u) / ps.slice, rec_blocks = (ns + 255u) / 256u;
        const bool small_lds = nbmax <= 8192u;             // direct scatter: 16 KB of packed counters are enough (c <= 13)
        const uint32_t scatter_passes = env_sp ? env_sp : (ns >= (1u << 19) ? 4u : ns >= (1u << 18) ? 2u : 1u);   // measured: profiles/r02_scatter_passes.txt
        const uint32_t *in = reinterpret_cast<const uint32_t *>(d_in) + (size_t)r0 * rec_words;
        const uint32_t split_g = last ? split_top : 0xffffffffu;          // earlier shards: one task order, one accumulate launch
        // what the accumulate / fold kernels read from the sort stage: one set per shard parity when the two overlap
        const size_t par = overlap ? (size_t)(sh & 1) : 0;
        auto *pts = reinterpret_cast<Aff<F> *>(static_cast<char *>(e->pts.p) + par * (size_t)ns_max * pt_bytes);
        PtL *ptl = limb_form ? reinterpret_cast<PtL *>(pts) : nullptr;
        void *limb_recs = (limb_form || g2_limb) ? static_cast<void *>(pts) : nullptr;      // PtL (G1) | PtL2 (G2) records in place of the affine points
        auto *taskoff = reinterpret_cast<uint32_t *>(e->taskoff.p) + par * (size_t)(pl.NB + 1);
        auto *entries = reinterpret_cast<uint32_t *>(e->entries.p) + par * (size_t)pl.W * ns_max;
        auto *tasks = reinterpret_cast<Task *>(e->tasks.p) + par * (size_t)pl.max_tasks;
        auto *perm = reinterpret_cast<uint32_t *>(e->perm.p) + par * (size_t)pl.max_tasks;
        uint32_t *split_small = reinterpret_cast<uint32_t *>(e->split_lists.p) + par * 2u * (size_t)pl.NB, *split_big = split_small + pl.NB;
        auto *blk = reinterpret_cast<uint32_t *>(e->scan_blk.p) + par * kScanBlkWords;
        uint32_t *lenhist = blk + kLenHist, *lenoff = blk + kLenOff, *ranges = blk + kRanges;
        uint32_t *taskbkt = two_level ? reinterpret_cast<uint32_t *>(e->taskbkt.p) + par * (size_t)pl.max_tasks : nullptr;
        auto *totals = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 16 + 16 * par);
        if (staged && !inline_copies) {
            if (!staged_copy.wait_shard(sh)) {
                fprintf(stderr, "[eip2537_hip] host-to-device copy of shard %d failed\n", sh);
                e->failed = true;
                return E_MEMORY_ERROR;
            }
            HIPCHK(hipStreamWaitEvent(ss, e->ev_copy[sh], 0));
        } else if (host_src) {
            // Measured (profiles/r03_h2d_chunks.txt): chunking ONE pipeline's copy with decode + histograms behind every chunk lost
            // (7.31 -> 7.51 .. 8.08 ms at 2^20: only 0.2 ms of work could follow a chunk); that code is gone.
            const hipError_t ce = hipMemcpyAsync(const_cast<uint32_t *>(in), static_cast<const char *>(host_src) + (size_t)r0 * rec_bytes, (size_t)ns * rec_bytes, hipMemcpyHostToDevice, ss);
            if (e->copy_gate) e->copy_gate->done(e->copy_turn);            // shards of one call on one device copy in shard order (api.hip); a pageable copy returns when its last chunk is staged
            HIPCHK(ce);
        }
        if (overlap && sh >= 2) HIPCHK(hipStreamWaitEvent(ss, e->ev_accdone[sh - 2], 0));      // this parity's set is free again
        // totals: [0] entries [1] tasks [2] lightly split [3] heavily split buckets -- written / cleared by k_msm_scan_top
        hipLaunchKernelGGL(k_msm_decode<F>, dim3(rec_blocks), dim3(256), 0, ss, in, ps, pts, limb_recs, digits, err, r0);
        if (!sort2 && small_lds) hipLaunchKernelGGL(k_msm_hist<4096u>, dim3(nslices, pl.W), dim3(1024), 0, ss, digits, ps, nslices, nbmax, hist16, 0u);
        else if (!sort2) hipLaunchKernelGGL(k_msm_hist<kLdsWords>, dim3(nslices, pl.W), dim3(1024), 0, ss, digits, ps, nslices, nbmax, hist16, 0u);
        const uint32_t scan_blocks = (pl.NB + 1023u) / 1024u;      // <= 1024 (c <= 16)
        if (sort2) {
            // a degenerate input (k_sort_coarse_scan raises `heavy`) leaves every bucket empty here; the host then re-runs the call with
            // the direct scatter (below).  Round 3 launched the direct-scatter kernels behind these with the flag as their condition:
            // three empty launches (17 us) in every shard of every ordinary call.
            uint32_t *wtotal = blk + kWTotal, *wbase = blk + kWBase;
            hipLaunchKernelGGL(k_sort_coarse_hist, dim3(nslices, pl.W), dim3(1024), 0, ss, digits, ps, nslices, hist16, heavy);
            hipLaunchKernelGGL(k_sort_coarse_scan, dim3(pl.W), dim3(1024), 0, ss, hist16, ps, nslices, wtotal, heavy);
            hipLaunchKernelGGL(k_sort_window_bases, dim3(1), dim3(64), 0, ss, wtotal, pl.W, wbase);
            if (ps.slice <= 16384u)
                hipLaunchKernelGGL(k_sort_coarse_scatter<16384u>, dim3(nslices, pl.W), dim3(1024), 0, ss, digits, ps, nslices, hist16, wbase, base, (const uint32_t *)heavy);
            else
                hipLaunchKernelGGL(k_sort_coarse_scatter<kSlice>, dim3(nslices, pl.W), dim3(1024), 0, ss, digits, ps, nslices, hist16, wbase, base, (const uint32_t *)heavy);
            hipLaunchKernelGGL(k_sort_fine, dim3(std::max(pl.B, pl.BT) >> kFineBits, pl.W), dim3(512), 0, ss, base, ps, nslices, hist16, wbase, entries, counts, (const uint32_t *)heavy);
        } else {
            hipLaunchKernelGGL(k_msm_slicescan, dim3((nbmax / 2u + 255u) / 256u, pl.W), dim3(256), 0, ss, hist16, ps, nslices, nbmax, base, counts);
        }
        hipLaunchKernelGGL(k_msm_scan_sums, dim3(scan_blocks), dim3(1024), 0, ss, counts, pl.NB, lshift, blk);
        hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(1024), 0, ss, blk, scan_blocks, pl.NB, taskoff, totals, lenhist);
        hipLaunchKernelGGL(k_msm_scan_apply, dim3(scan_blocks), dim3(1024), 0, ss, counts, pl.NB, lshift, blk, offsets, taskoff);
        if (!sort2 && small_lds)
            hipLaunchKernelGGL(k_msm_scatter<4096u>, dim3(8u * nslices * (((uint32_t)pl.W + 7u) / 8u)), dim3(1024), 0, ss, digits, ps, nslices, nbmax, base, offsets, entries, scatter_passes);
        else if (!sort2)
            hipLaunchKernelGGL(k_msm_scatter<kLdsWords>, dim3(8u * nslices * (((uint32_t)pl.W + 7u) / 8u)), dim3(1024), 0, ss, digits, ps, nslices, nbmax, base, offsets, entries, scatter_passes);
        hipLaunchKernelGGL(k_msm_tasks, dim3((pl.NB + kTaskItems - 1u) / kTaskItems), dim3(256), 0, ss, counts, offsets, taskoff, pl.NB, lshift, tasks,
                           split_small, split_big, totals + 2, lenhist, gshift, split_g, taskbkt, bacc, first_shard);
        const uint32_t task_blocks = (ps.max_tasks + 255u) / 256u;
        hipLaunchKernelGGL(k_msm_task_scan, dim3(1), dim3(64), 0, ss, lenhist, lenoff, ranges);
        hipLaunchKernelGGL(k_msm_task_perm, dim3((ps.max_tasks + kTaskItems - 1u) / kTaskItems), dim3(256), 0, ss, tasks, totals, lenoff, perm, gshift, (const uint32_t *)taskoff, split_g);
        if (overlap) {
            HIPCHK(hipEventRecord(e->ev_sorted[sh], ss));
            HIPCHK(hipStreamWaitEvent(s, e->ev_sorted[sh], 0));
        }
        if (last) HIPCHK(hipEventRecord(e->ev_a, s));
        if (two_level) {
            int st2 = E_SUCCESS;
            st2 = launch_two_level_or_shard(e, ps, ns, lshift, task_blocks, red_blocks, split_g, last, ptl, entries, tasks, perm, totals, ranges, partial, taskoff,
                                            split_small, split_big, taskbkt, bacc, reinterpret_cast<Xyzz<FpL> *>(e->rcsum.p), winout, first_shard);
            if (st2) return st2;
        } else {
            // c <= 13 plans: the accumulate is chain-bound, two lanes per task (G1 2^16 1.15 -> 1.07 ms, 2^17 1.67 -> 1.48 ms)
            launch_accum(s, task_blocks, pl.c <= 13, pts, limb_recs, entries, tasks, perm, totals, partial);
            HIPCHK(hipEventRecord(e->ev_b, s));
            launch_fold_small(s, four, limb_form || g2_limb, partial, taskoff, split_small, totals + 2);
            launch_fold_big(s, limb_form || g2_limb, partial, taskoff, split_big, totals + 2);
            launch_reduce(s, red_blocks, four, limb_form || g2_limb, partial, taskoff, pl, rg, winout, two_level_p ? &rcp : nullptr, reinterpret_cast<Xyzz<FpL> *>(e->rcsum.p));
        }
        if (overlap) HIPCHK(hipEventRecord(e->ev_accdone[sh], s));
    }
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());

    // what comes back: first-error word | `heavy` flag | the window sums -- into the slot's pinned buffer (two direct copies, no staging)
    HIPCHK(e->need_pinned(16 + nwin_out * sizeof(Xyzz<F>)));
    unsigned long long *herr2 = static_cast<unsigned long long *>(e->pinned);
    Xyzz<F> *hw = reinterpret_cast<Xyzz<F> *>(static_cast<char *>(e->pinned) + 16);
    StreamDrain drain{s};
    HIPCHK(hipMemcpyAsync(herr2, err, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hw, winout + (dev_winsum ? red_blocks : 0u), nwin_out * sizeof(Xyzz<F>), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    drain.armed = false;
    const unsigned long long herr = herr2[0];
    staged_copy.finish();                       // (the helper has long finished: every copy event was waited for above)
    if (sort2 && (uint32_t)herr2[1] != 0u && herr == ~0ull) {
        // degenerate input (a partition above kHeavyFactor times its window's mean, e.g. all scalars equal): the partitioned sort
        // stood down on the device and the buckets above are empty.  The records are all in HBM by now: run the call again with the
        // direct scatter, which spreads a heavy bucket over its 32 768-record slices (profiles/r04_degenerate_inputs.txt).
        return msm_device_t<F>(e, d_in, n, partial_words, c_override, true);
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_a) == hipSuccess) e->last_aux_ms[0] = ms;     // decode .. task order (staged call: everything before the last shard's accumulate)
    if (hipEventElapsedTime(&ms, e->ev_b, e->ev_stop) == hipSuccess) e->last_aux_ms[1] = ms;      // fold + bucket reduce
    if (herr != ~0ull) return (int)(herr & 7ull);

    // Horner over windows on the host (W * c doublings + a handful of additions); G2: the accumulator stays in one IFMA vector (ifma_horner.h)
    HornerAcc<F> h;
    if (two_level) {
        // window sum = 256 R_w + C_w: two half-windows of 8 bits each (C_w arrives as its lower and upper columns); the top window is its
        // two halves added
        for (int w = 0; w <= pl.W; w++) hw[3 * (size_t)w + 1] = add(hw[3 * (size_t)w + 1], hw[3 * (size_t)w + 2]);
        hw[3 * (size_t)(pl.W - 1)] = add(hw[3 * (size_t)(pl.W - 1)], hw[3 * (size_t)pl.W]);
        hw[3 * (size_t)(pl.W - 1) + 1] = add(hw[3 * (size_t)(pl.W - 1) + 1], hw[3 * (size_t)pl.W + 1]);
        for (int w = pl.W - 1; w >= 0; w--)
            for (int hf = 0; hf < 2; hf++) {
                h.dbl_n(pl.c / 2);
                h.add(hw[3 * w + hf]);
            }
    } else if (two_level_p) {
        // window sum = 64 R_w + C_w: a 7-bit and a 6-bit half-window
        for (int w = pl.W - 1; w >= 0; w--) {
            h.dbl_n(pl.c - (int)rcp.logC);
            h.add(hw[2 * (size_t)w]);
            h.dbl_n((int)rcp.logC);
            h.add(hw[2 * (size_t)w + 1]);
        }
    } else if (dev_winsum) {
        for (int w = pl.W - 1; w >= 0; w--) {
            h.dbl_n(pl.c);
            h.add(hw[(size_t)w]);
        }
    } else
    for (int w = pl.W - 1; w >= 0; w--) {
        h.dbl_n(pl.c);
        const uint32_t nb = w == pl.W - 1 ? rg.bt : rg.bn, b0 = (uint32_t)w * rg.bn;
        for (uint32_t b = 0; b < nb; b++) h.add(hw[b0 + b]);
    }
    const Xyzz<F> acc = h.result();
    memcpy(partial_words, &acc, sizeof acc);
    return E_SUCCESS;
}

// ---- coalesced batch of small MSMs ------------------------------------------------
// M concurrent small calls (each below 2048 records, i.e. the c = 8 plan: 32 windows of 8 bits, 128
// signed buckets, 256 in the unsigned top window) run as ONE pipeline over their concatenated records.
// A (window, call) pair owns kBatchBmax = 256 consecutive bucket ids,
//        bucket id = (window * M + call) * 256 + (bucket value - 1),
// so for the counting sort, the task split and the accumulate the batch simply is one MSM whose windows
// have M * 256 buckets (the kernels above run unchanged on that virtual plan); only the two ends know
// about calls: the decode (record -> call by binary search over the call offsets, a first-error word
// per call so that one bad call fails alone) and the reduce (one wave per (window, call): 16 four-lane
// groups of 16 buckets, wavefront tree, no cross-call sums).  The host gets 32 window sums per call
// and each caller finishes its own Horner / inversion / encoding on its own thread.
static constexpr uint32_t kBatchBmax = 256;
static constexpr int kBatchC = 8, kBatchW = 32;
template <class F>
__global__ void __launch_bounds__(256)
k_msm_decode_batch(const uint32_t *__restrict__ in, MsmPlan real, uint32_t ntotal, const uint32_t *__restrict__ coff, int M,
                   Aff<F> *__restrict__ pts, PtL *__restrict__ ptl, uint32_t *__restrict__ digits, unsigned long long *err) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= ntotal) return;
    int lo = 0, hi = M;                                   // call j with coff[j] <= i < coff[j + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (coff[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t j = (uint32_t)lo;
    bool live = false;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t *w = in + (size_t)i * Wire<F>::kMsmRecWords;
    int st;
    if (std::is_same<F, Fp>::value && ptl) {                  // uniform: limb records straight from the wire (G1)
        st = decode_point_limbs(&ptl[i], w, live);
    } else {
        Aff<F> a;
        st = decode_point_inl<F>(a, w);
        if (st == E_SUCCESS && !is_inf(a)) {
            pts[i] = a;
            live = true;
        }
    }
    if (st != E_SUCCESS) atomicMin(&err[j], ((unsigned long long)(i - coff[j]) << 3) | (unsigned long long)st);
    else if (live) decode_scalar(k, w + Wire<F>::kPointWords);
    int wi = 0;
    for_each_digit(k, real, [&](uint32_t g, uint32_t ng, bool nz) {
        const uint32_t v = g - (uint32_t)wi * real.B + 1u;                // bucket value 1..nb_w of the call's own plan
        digits[(size_t)wi * ntotal + i] = (live && nz) ? (((j * kBatchBmax + v) << 1) | ng) : 0u;
        wi++;
    });
}
// WPU waves per (window, call) unit: 16 * WPU four-lane groups of 16 / WPU buckets each, wavefront tree,
// and for WPU = 4 an LDS step across the block's waves.  Small batches take WPU = 4 (a chain of 8
// running-sum additions instead of 32) while their 128 * M waves still fit the chip in one round.
template <class F, int WPU, class T = typename AccumField<F>::T>
__global__ void __launch_bounds__(256, 1)
k_msm_reduce_batch(const Xyzz<F> *__restrict__ partial_, const uint32_t *__restrict__ taskoff, uint32_t units,
                   Xyzz<F> *__restrict__ winout) {
    const Xyzz<T> *__restrict__ partial = reinterpret_cast<const Xyzz<T> *>(partial_);
    claim_whole_simd();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 3, gb = lane & ~3;
    const uint32_t unit = WPU == 4 ? blockIdx.x : blockIdx.x * 4u + (uint32_t)wave;
    if (unit >= units) return;                                 // uniform in the wave (WPU = 4: in the block)
    constexpr uint32_t S = kBatchBmax / (16u * WPU);
    const uint32_t grp = (WPU == 4 ? (uint32_t)wave * 16u : 0u) + (uint32_t)(lane >> 2);
    const uint32_t lo = grp * S, base = unit * kBatchBmax;
    Xyzz<T> R = xyzz_inf<T>(), Q = xyzz_inf<T>();
    for (uint32_t v = lo + S; v > lo; v--) {
        const uint32_t g = base + v - 1u;
        const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
        if (t1 > t0) R = add4(R, partial[t0], r, gb);          // multi-task buckets were folded into slot t0
        Q = add4(Q, R, r, gb);
    }
    Xyzz<T> C = add4(Q, small_mul4(R, lo, r, gb), r, gb);      // sum_{v in (lo, lo + S]} v * B_v
    for (int off = 4; off < 64; off <<= 1) {
        Xyzz<T> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 4) C = add4(C, o, r, gb);
    }
    if (WPU == 4) {
        __shared__ Xyzz<T> sm[4];
        if (lane == 0) sm[wave] = C;
        __syncthreads();
        if (wave == 0 && lane < 4) {
            for (int k = 1; k < 4; k++) C = add4(C, sm[k], r, 0);
            if (lane == 0) store_canon<F, T>(&winout[unit], C);
        }
    } else if (lane == 0) {
        store_canon<F, T>(&winout[unit], C);
    }
}

// The same over Fp2 with the 8-lane component-split operations (lanes.h: add8c, small_mul8c): the 4-lane form kept whole
// Fp2 points replicated per lane (512 registers + 640-704 B of scratch per lane).  A wave holds 8 groups, so a unit is
// 8 * WPU runs of 256 / (8 WPU) buckets.
template <int WPU>
__global__ void __launch_bounds__(256, 1)
k_msm_reduce_batch8c(const Xyzz<Fp2> *__restrict__ partial, const uint32_t *__restrict__ taskoff, uint32_t units,
                     Xyzz<Fp2> *__restrict__ winout) {
    claim_whole_simd();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane & 7, gb = lane & ~7, q = sl & 1;
    const uint32_t unit = WPU == 4 ? blockIdx.x : blockIdx.x * 4u + (uint32_t)wave;
    if (unit >= units) return;                                 // uniform in the wave (WPU = 4: in the block)
    const PairProd8 prod(lane, sl, gb);
    constexpr uint32_t S = kBatchBmax / (8u * WPU);
    const uint32_t grp = (WPU == 4 ? (uint32_t)wave * 8u : 0u) + (uint32_t)(lane >> 3);
    const uint32_t lo = grp * S, base = unit * kBatchBmax;
    const Xyzz<FpI> inf = xyzz_inf<FpI>();
    Xyzz<FpI> R = inf, Q = inf;
    for (uint32_t v = lo + S; v > lo; v--) {
        const uint32_t g = base + v - 1u;
        const uint32_t t0 = taskoff[g], t1 = taskoff[g + 1];
        if (t1 > t0) R = add8c(R, component_of(partial[t0], q), prod);      // multi-task buckets were folded into slot t0
        Q = add8c(Q, R, prod);
    }
    Xyzz<FpI> C = add8c(Q, small_mul8c(R, lo, prod), prod);    // sum_{v in (lo, lo + S]} v * B_v
    for (int off = 8; off < 64; off <<= 1) {
        Xyzz<FpI> o = shfl_from(C, (lane + off) & 63);
        if ((lane & (2 * off - 1)) < 8) C = add8c(C, o, prod);
    }
    if (WPU == 4) {
        __shared__ Xyzz<FpI> sm[4][2];              // [wave][component]
        if (lane < 2) sm[wave][lane] = C;
        __syncthreads();
        if (wave == 0 && lane < 8) {
            for (int k = 1; k < 4; k++) C = add8c(C, sm[k][q], prod);
            if (lane < 2) store_component(&winout[unit], C, q);
        }
    } else if (lane < 2) {
        store_component(&winout[unit], C, q);
    }
}
static void launch_reduce_batch(hipStream_t s, uint32_t units, bool limb, const Xyzz<Fp> *partial, const uint32_t *taskoff, Xyzz<Fp> *winout) {
    const bool wide = units * 4u <= 1024u;       // one block per unit while 4 waves per unit fit one round of one wave per SIMD
    const dim3 grid(wide ? units : (units + 3u) / 4u);
    if (limb && wide) hipLaunchKernelGGL((k_msm_reduce_batch<Fp, 4, FpL>), grid, dim3(256), 0, s, partial, taskoff, units, winout);
    else if (limb) hipLaunchKernelGGL((k_msm_reduce_batch<Fp, 1, FpL>), grid, dim3(256), 0, s, partial, taskoff, units, winout);
    else if (wide) hipLaunchKernelGGL((k_msm_reduce_batch<Fp, 4>), grid, dim3(256), 0, s, partial, taskoff, units, winout);
    else hipLaunchKernelGGL((k_msm_reduce_batch<Fp, 1>), grid, dim3(256), 0, s, partial, taskoff, units, winout);
}
static void launch_reduce_batch(hipStream_t s, uint32_t units, bool, const Xyzz<Fp2> *partial, const uint32_t *taskoff, Xyzz<Fp2> *winout) {
    // (round 4: the replicated 4-lane form k_msm_reduce_batch<Fp2, *> of round 2 -- 512 registers + 800 B of scratch per lane, kept behind
    // an A/B switch in round 3 -- is no longer instantiated)
    if (units * 4u <= 1024u) hipLaunchKernelGGL(k_msm_reduce_batch8c<4>, dim3(units), dim3(256), 0, s, partial, taskoff, units, winout);
    else hipLaunchKernelGGL(k_msm_reduce_batch8c<1>, dim3((units + 3u) / 4u), dim3(256), 0, s, partial, taskoff, units, winout);
}

// d_in: the M calls' records back to back in HBM; coff[0..M]: record offsets of the calls (host memory).
// Writes rc[j] (0 or the EIP2537 code of call j's lowest bad record) and, for the good calls, their
// kBatchW window sums to wins[j * kBatchW ...].  Returns non-zero only for a HIP failure (every call fails).
template <class F>
static int msm_batch_device_t(Engine *e, const void *d_in, const uint32_t *coff, int M, Xyzz<F> *wins, int *rc) {
    const size_t n = coff[M];
    if (M < 1 || M > 256 || n == 0 || n >= (1ull << 24)) return E_MEMORY_ERROR;
    MsmPlan real = msm_make_plan(1024, kBatchC, ReduceCfg<F>::kFourLane);          // the plan every call below 2048 records has
    if (real.c != kBatchC || real.W != kBatchW || real.B != 128u || real.BT != kBatchBmax) return E_MEMORY_ERROR;
    MsmPlan pl = real;                                        // the virtual plan of the whole batch
    pl.n = (uint32_t)n;
    pl.B = pl.BT = (uint32_t)M * kBatchBmax;
    pl.NB = (uint32_t)pl.W * pl.B;
    pl.max_entries = (uint64_t)n * pl.W;
    const uint32_t lshift = kMinTaskShift;                    // tasks of <= 16 entries: these plans are chain-bound
    const uint32_t gshift = 0;
    pl.L = 1u << lshift;
    pl.max_tasks = (uint32_t)(pl.NB + pl.max_entries / pl.L + 1);
    const uint32_t units = (uint32_t)pl.W * (uint32_t)M;
    pl.slice = msm_slice_for((uint32_t)n, pl.W, false);
    const uint32_t nslices = (uint32_t)((n + pl.slice - 1) / pl.slice);
    const uint32_t nbmax = pl.B;
    if (nbmax > 2u * kLdsWords) return E_MEMORY_ERROR;         // LDS histogram: 65536 packed counters
    // G1: the whole batch pipeline in limb form (limb30.h), like the single-call plans of the same size
    static const bool env_limb = [] { const char *v = getenv("EIP2537_LIMB_FORM"); return !v || atoi(v) != 0; }();
    const bool limb_form = !ReduceCfg<F>::kFourLane && env_limb;
    HIPCHK(e->pts.reserve(n * (limb_form ? sizeof(PtL) : sizeof(Aff<F>))));
    HIPCHK(e->counts.reserve((size_t)pl.NB * 4));
    HIPCHK(e->offsets.reserve((size_t)pl.NB * 4));
    HIPCHK(e->digits.reserve((size_t)pl.W * n * 4));
    HIPCHK(e->hist16.reserve((size_t)pl.W * nslices * (nbmax / 2) * 4));
    HIPCHK(e->slice_base.reserve((size_t)pl.W * nslices * nbmax * 4));
    HIPCHK(e->taskoff.reserve((size_t)(pl.NB + 1) * 4));
    HIPCHK(e->entries.reserve(pl.max_entries * 4));
    HIPCHK(e->tasks.reserve((size_t)pl.max_tasks * sizeof(Task)));
    HIPCHK(e->partial.reserve((size_t)pl.max_tasks * (limb_form ? sizeof(Xyzz<FpL>) : sizeof(Xyzz<F>))));
    HIPCHK(e->winout.reserve((size_t)units * sizeof(Xyzz<F>)));
    HIPCHK(e->misc.reserve(64 + (size_t)(M + 1) * 4 + (size_t)M * 8));
    HIPCHK(e->scan_blk.reserve(kScanBlkWords * 4));
    HIPCHK(e->perm.reserve((size_t)pl.max_tasks * 4));
    HIPCHK(e->split_lists.reserve((size_t)pl.NB * 8));
    if ((pl.NB + 1023u) / 1024u > 1024u) return E_MEMORY_ERROR;

    const uint32_t scatter_passes = 1u;
    hipStream_t s = e->stream;
    auto *totals = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 16);
    auto *err = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(e->misc.p) + 64);        // [M]
    auto *d_coff = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(e->misc.p) + 64 + (size_t)M * 8);   // [M + 1]
    HIPCHK(hipMemsetAsync(err, 0xFF, (size_t)M * 8, s));
    HIPCHK(hipMemcpyAsync(d_coff, coff, (size_t)(M + 1) * 4, hipMemcpyHostToDevice, s));
    const uint32_t *in = reinterpret_cast<const uint32_t *>(d_in);
    auto *pts = reinterpret_cast<Aff<F> *>(e->pts.p);
    auto *digits = reinterpret_cast<uint32_t *>(e->digits.p);
    auto *counts = reinterpret_cast<uint32_t *>(e->counts.p);
    auto *offsets = reinterpret_cast<uint32_t *>(e->offsets.p);
    auto *hist16 = reinterpret_cast<uint32_t *>(e->hist16.p);
    auto *base = reinterpret_cast<uint32_t *>(e->slice_base.p);
    auto *taskoff = reinterpret_cast<uint32_t *>(e->taskoff.p);
    auto *entries = reinterpret_cast<uint32_t *>(e->entries.p);
    auto *tasks = reinterpret_cast<Task *>(e->tasks.p);
    auto *partial = reinterpret_cast<Xyzz<F> *>(e->partial.p);
    auto *winout = reinterpret_cast<Xyzz<F> *>(e->winout.p);
    auto *blk = reinterpret_cast<uint32_t *>(e->scan_blk.p);
    uint32_t *lenhist = blk + kLenHist, *lenoff = blk + kLenOff, *ranges = blk + kRanges;
    auto *perm = reinterpret_cast<uint32_t *>(e->perm.p);
    uint32_t *split_small = reinterpret_cast<uint32_t *>(e->split_lists.p), *split_big = split_small + pl.NB;
    {
        LastPlan lp{};
        if (ReduceCfg<F>::kFourLane) snprintf(lp.kernel, sizeof lp.kernel, "k_msm_accum2c");
        else if (limb_form) snprintf(lp.kernel, sizeof lp.kernel, "k_msm_accum2_l");
        else snprintf(lp.kernel, sizeof lp.kernel, "k_msm_accum2<%s>", ReduceCfg<F>::kName);
        lp.c = pl.c; lp.windows = pl.W; lp.lanes = 2; lp.units = (uint32_t)n; lp.buckets = pl.NB; lp.shards = 1;
        e->last_plan = lp;
    }
    HIPCHK(hipEventRecord(e->ev_start, s));
    PtL *ptl = limb_form ? reinterpret_cast<PtL *>(e->pts.p) : nullptr;
    hipLaunchKernelGGL(k_msm_decode_batch<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, real, (uint32_t)n, d_coff, M, pts, ptl, digits, err);
    if (nbmax <= 8192u) hipLaunchKernelGGL(k_msm_hist<4096u>, dim3(nslices, pl.W), dim3(1024), 0, s, digits, pl, nslices, nbmax, hist16, 0u);
    else hipLaunchKernelGGL(k_msm_hist<kLdsWords>, dim3(nslices, pl.W), dim3(1024), 0, s, digits, pl, nslices, nbmax, hist16, 0u);
    hipLaunchKernelGGL(k_msm_slicescan, dim3((nbmax / 2u + 255u) / 256u, pl.W), dim3(256), 0, s, hist16, pl, nslices, nbmax, base, counts);
    const uint32_t scan_blocks = (pl.NB + 1023u) / 1024u;
    hipLaunchKernelGGL(k_msm_scan_sums, dim3(scan_blocks), dim3(1024), 0, s, counts, pl.NB, lshift, blk);
    hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(1024), 0, s, blk, scan_blocks, pl.NB, taskoff, totals, lenhist);
    hipLaunchKernelGGL(k_msm_scan_apply, dim3(scan_blocks), dim3(1024), 0, s, counts, pl.NB, lshift, blk, offsets, taskoff);
    if (nbmax <= 8192u) hipLaunchKernelGGL(k_msm_scatter<4096u>, dim3(8u * nslices * (((uint32_t)pl.W + 7u) / 8u)), dim3(1024), 0, s, digits, pl, nslices, nbmax, base, offsets, entries, scatter_passes);
    else hipLaunchKernelGGL(k_msm_scatter<kLdsWords>, dim3(8u * nslices * (((uint32_t)pl.W + 7u) / 8u)), dim3(1024), 0, s, digits, pl, nslices, nbmax, base, offsets, entries, scatter_passes);
    hipLaunchKernelGGL(k_msm_tasks, dim3((pl.NB + kTaskItems - 1u) / kTaskItems), dim3(256), 0, s, counts, offsets, taskoff, pl.NB, lshift, tasks,
                       split_small, split_big, totals + 2, lenhist, gshift, 0xffffffffu);
    hipLaunchKernelGGL(k_msm_task_scan, dim3(1), dim3(64), 0, s, lenhist, lenoff, ranges);
    const uint32_t task_blocks = (pl.max_tasks + 255u) / 256u;
    hipLaunchKernelGGL(k_msm_task_perm, dim3((pl.max_tasks + kTaskItems - 1u) / kTaskItems), dim3(256), 0, s, tasks, totals, lenoff, perm, gshift, (const uint32_t *)taskoff, 0xffffffffu);
    HIPCHK(hipEventRecord(e->ev_a, s));
    launch_accum(s, task_blocks, true, pts, ptl, entries, tasks, perm, totals, partial);          // two lanes per task
    HIPCHK(hipEventRecord(e->ev_b, s));
    launch_fold_small(s, true, limb_form, partial, taskoff, split_small, totals + 2);
    launch_fold_big(s, limb_form, partial, taskoff, split_big, totals + 2);
    launch_reduce_batch(s, units, limb_form, partial, taskoff, winout);
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());

    std::vector<unsigned long long> herr((size_t)M);
    std::vector<Xyzz<F>> hw(units);
    StreamDrain drain{s};
    HIPCHK(hipMemcpyAsync(herr.data(), err, (size_t)M * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hw.data(), winout, (size_t)units * sizeof(Xyzz<F>), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    drain.armed = false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    for (int j = 0; j < M; j++) {
        rc[j] = herr[(size_t)j] != ~0ull ? (int)(herr[(size_t)j] & 7ull) : E_SUCCESS;
        for (int w = 0; w < kBatchW; w++) wins[(size_t)j * kBatchW + w] = hw[(size_t)w * (uint32_t)M + (uint32_t)j];
    }
    return E_SUCCESS;
}
int msm_g1_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *wins_words, int *rc) {
    return msm_batch_device_t<Fp>(e, d_in, coff, M, reinterpret_cast<Xyzz<Fp> *>(wins_words), rc);
}
int msm_g2_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *wins_words, int *rc) {
    return msm_batch_device_t<Fp2>(e, d_in, coff, M, reinterpret_cast<Xyzz<Fp2> *>(wins_words), rc);
}

int msm_g1_device(Engine *e, const void *d_in, size_t n, uint32_t *partial_words, int c_override) {
    return msm_device_t<Fp>(e, d_in, n, partial_words, c_override);
}
int msm_g2_device(Engine *e, const void *d_in, size_t n, uint32_t *partial_words, int c_override) {
    return msm_device_t<Fp2>(e, d_in, n, partial_words, c_override);
}

}  // namespace eip
