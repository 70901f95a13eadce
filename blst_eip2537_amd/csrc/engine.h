// Internal interface between the translation units of libeip2537_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace eip {

// Grow-only device allocation, owned by the engine context.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes);
    void release();
};

struct Engine {
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_a = nullptr, ev_b = nullptr, ev_j2 = nullptr, ev_j3 = nullptr;
    // staging + per-call workspace (grow-only)
    DevBuf input;          // H2D copy of a host caller's records
    DevBuf misc;           // first-error word, scan totals, split-bucket counters
    // MSM
    DevBuf pts;            // decoded affine points, AoS
    DevBuf digits;         // [window][record] bucket value / sign
    DevBuf hist16;         // [window][slice][bucket] packed 16-bit slice histograms
    DevBuf slice_base;     // [window][slice][bucket] exclusive prefix over the slices
    DevBuf counts, offsets, taskoff;   // per bucket: entries, first entry, first task
    DevBuf scan_blk;       // block totals of the bucket scan + task-length histogram
    DevBuf entries;        // (record << 1 | sign), sorted by bucket
    DevBuf tasks, perm;    // <= L-entry runs of one bucket; tasks ordered by length
    DevBuf split_lists;    // buckets split into several tasks: lightly | heavily
    DevBuf partial;        // one XYZZ point per task        (pairing: the 68 x k line records)
    DevBuf winout;         // per (window, reduce block) sums (pairing: per-block / per-step Fp12 products)
    // last-call kernel timing (ms), filled when timing is enabled
    float last_kernel_ms = 0.f;   // whole device pipeline of the last call
    float last_accum_ms = 0.f;    // dominant kernel of the last call (k_msm_accum / k_pair_miller)
};

// Window plan for one MSM
struct MsmPlan {
    uint32_t n;        // records
    int c;             // window bits
    int W;             // windows
    int topbits;       // bits in the top (unsigned) window
    uint32_t B;        // buckets per signed window = 2^(c-1)
    uint32_t BT;       // buckets in the top window = 2^topbits
    uint32_t NB;       // total buckets
    uint32_t L;        // max entries per accumulate task
    uint32_t S;        // buckets per reduce segment
    uint32_t max_tasks;
    uint64_t max_entries;
};
MsmPlan msm_make_plan(uint32_t n, int c_override, bool g2);

// Device pipelines.  `d_in` is the EIP-encoded record stream resident in HBM (4-byte aligned).
// On success returns 0 and writes the projective partial sum (XYZZ, Montgomery limbs: 48 words
// for G1, 96 for G2) to host memory `partial_words`.  On a data error returns the EIP2537 code of
// the lowest-index bad record.
int msm_g1_device(Engine *e, const void *d_in, size_t n, uint32_t *partial_words, int c_override);
int msm_g2_device(Engine *e, const void *d_in, size_t n, uint32_t *partial_words, int c_override);

// Pairing: on success returns 0 and writes the product of the Miller loops (before the final
// exponentiation) as 144 Montgomery words to `ml_words`.
int pairing_device(Engine *e, const void *d_in, size_t k, uint32_t *ml_words);

}  // namespace eip
