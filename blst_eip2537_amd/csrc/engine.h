// Internal interface between the translation units of libeip2537_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace eip {

// Grow-only device allocation, owned by the engine context.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool fresh = false;    // set by reserve() when it allocated: the contents are undefined (a consumer that keeps state in the buffer re-initialises it and clears the flag)
    hipError_t reserve(size_t bytes);
    void release();
};

// What the last device pipeline on an engine actually ran (reported through
// eip2537_hip_last_plan so that bench.py labels its roofline from the library, not from a copy of
// the dispatch thresholds).
struct LastPlan {
    char kernel[48];   // dominant kernel, as rocprofv3 names it
    int c;             // MSM window bits (0 for a pairing batch)
    int windows;       // MSM windows (pairing: Miller steps)
    int lanes;         // lanes per task / pair of the dominant kernel
    uint32_t units;    // records / pairs of the launch
    uint32_t buckets;  // MSM buckets (0 for a pairing batch)
    int shards;        // record shards the call was staged in (1: one copy / resident input)
};

// Compute units / SIMDs of a device, read once per device from hipGetDeviceProperties (msm.hip).  The chain-bound kernels size
// their grids to "one wave per SIMD"; nothing assumes the full 256-CU part any more (a CPX partition or a CU mask reports
// fewer).  $EIP2537_HIP_CUS overrides the count (tests).
struct ChipShape { uint32_t cus, simds; };
ChipShape chip_shape(int device);

// Order of the host -> device copies of the shards of ONE host-input call that share a device (api.hip, msm_host_abi):
// shard s stages its records only after shard s - 1 has handed its own to the copy engine, so that the shards' pipelines
// run one behind the other -- shard s computes while shard s + 1 copies -- instead of sharing the PCIe link.
// (Round 4: G2 only; G1 calls are staged inside ONE pipeline, StagedCopy below.  The ordering relies on a PAGEABLE hipMemcpyAsync
// returning only when its last byte is staged; from a pinned / registered caller buffer the copy returns at once and the shards' copies then
// share the link -- slower, never wrong: every shard's kernels are ordered behind its own copy by its stream.)
struct CopyGate {
    std::mutex m;
    std::condition_variable cv;
    int turn = 0;
    void wait_turn(int s) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return turn >= s; });
    }
    void done(int s) {                             // idempotent; also called when a shard fails before its copy
        {
            std::lock_guard<std::mutex> lk(m);
            if (turn < s + 1) turn = s + 1;
        }
        cv.notify_all();
    }
};

// One persistent helper thread per engine slot, started by the slot's first staged call (round 4: no thread is created per
// call).  run() hands it one job; wait() returns when that job has finished.  The owner of the slot is the only caller.
class Helper {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || (busy && job); });
            if (quit) return;
            std::function<void()> f;
            f.swap(job);
            lk.unlock();
            f();
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
public:
    Helper() = default;
    Helper(const Helper &) = delete;
    Helper &operator=(const Helper &) = delete;
    bool run(std::function<void()> f) {                // false: no thread could be started (the caller does the work itself)
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) {
            try { th = std::thread([this] { loop(); }); } catch (...) { return false; }
        }
        cv.wait(lk, [&] { return !busy; });
        job = std::move(f);
        busy = true;
        cv.notify_all();
        return true;
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy; });
    }
    ~Helper() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// Record ranges of a host-input call that is staged shard by shard (api.hip decides the cut, msm.hip runs it): shard s is
// records [bound[s], bound[s + 1]).  k <= 1: one copy.
struct ShardFeed {
    static constexpr int kMax = 8;
    int k = 0;
    uint32_t bound[kMax + 1] = {};
};

// The copies of a staged call: the slot's helper thread hands the shards of the caller's buffer to `copy_stream` one behind the
// other (a pageable hipMemcpyAsync keeps its host thread until the last byte is staged, so the link never idles between shards
// and never carries two at once) and records an event behind each; the launching thread waits for "shard s handed over" and
// then orders that shard's kernels behind the event.  finish() returns when the helper no longer touches the caller's buffer:
// every exit of the pipeline, early error returns included, passes through it (destructor).
struct StagedCopy {
    std::mutex m;
    std::condition_variable cv;
    int posted = 0;
    bool failed = false;
    Helper *helper = nullptr;
    bool start(Helper &h, int device, hipStream_t copy_stream, hipEvent_t *events, const ShardFeed &f, size_t rec_bytes, void *dst, const void *src) {
        helper = &h;
        const ShardFeed feed = f;
        const bool ok = h.run([this, device, copy_stream, events, feed, rec_bytes, dst, src] {
            bool bad = hipSetDevice(device) != hipSuccess;
            for (int s = 0; s < feed.k; s++) {
                if (!bad) {
                    const size_t off = (size_t)feed.bound[s] * rec_bytes, len = (size_t)(feed.bound[s + 1] - feed.bound[s]) * rec_bytes;
                    bad = hipMemcpyAsync(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
                          hipEventRecord(events[s], copy_stream) != hipSuccess;
                }
                {
                    std::lock_guard<std::mutex> lk(m);
                    posted = s + 1;
                    failed = failed || bad;
                }
                cv.notify_all();
            }
            // a pinned / registered caller buffer makes the copies truly asynchronous: the buffer must not be read after the call
            // returns, so the helper only finishes when the copy stream has drained
            if (hipStreamSynchronize(copy_stream) != hipSuccess) {
                std::lock_guard<std::mutex> lk(m);
                failed = true;
            }
        });
        if (!ok) helper = nullptr;
        return ok;
    }
    bool wait_shard(int s) {                          // false: a copy failed
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return posted > s; });
        return !failed;
    }
    void finish() {
        if (helper) helper->wait();
        helper = nullptr;
    }
    ~StagedCopy() { finish(); }
};

struct Engine {
    bool ready = false;
    bool failed = false;   // a HIP call failed mid-pipeline: the slot is drained and rebuilt on release
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_a = nullptr, ev_b = nullptr, ev_j2 = nullptr, ev_j3 = nullptr, ev_c = nullptr;
    hipEvent_t ev_copy[ShardFeed::kMax] = {};      // behind the copy of every shard of a staged host-input call (stream2)
    hipEvent_t ev_sorted[ShardFeed::kMax] = {}, ev_accdone[ShardFeed::kMax] = {};      // staged call: sort stage (stream3) <-> accumulate (stream) of a shard
    Helper helper;                                 // stages those copies
    ShardFeed feed;                                // set with host_src by a staged call (k > 1)
    // staging + per-call workspace (grow-only)
    DevBuf input;          // H2D copy of a host caller's records
    CopyGate *copy_gate = nullptr;    // set with host_src by a shard of a pipelined host-input call
    int copy_turn = 0;
    const void *host_src = nullptr;   // set by a host-input MSM call: the pipeline stages `input` from here itself, in chunks (msm.hip)
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
    DevBuf bacc;           // G1 c = 16 plans: one limb-form XYZZ running sum per bucket, over all record shards of the call
    DevBuf taskbkt;        // task -> bucket << 1 | first task of its bucket
    DevBuf rcsum;          // row / column sums of the two-level reduce
    DevBuf winout;         // per (window, reduce block) sums (pairing: per-block / per-step Fp12 products)
    // last-call kernel timing (ms), filled when timing is enabled
    float last_kernel_ms = 0.f;   // whole device pipeline of the last call
    float last_accum_ms = 0.f;    // dominant kernel of the last call (named in last_plan.kernel)
    float last_aux_ms[2] = {0.f, 0.f};   // pairing: G1 membership kernel, line products (fold + tree2); MSM: sort stage (decode .. task order), fold + reduce
    LastPlan last_plan{};

    // Pinned host memory for the few KB every call brings back (error word, window sums / Miller products): a device-to-host copy
    // into pageable memory goes through the runtime's own staging buffer and a second host copy (round 4)
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    hipError_t need_pinned(size_t bytes) {
        if (bytes <= pinned_cap) return hipSuccess;
        if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; pinned_cap = 0; }
        const size_t want = bytes < (size_t)65536 ? (size_t)65536 : bytes + bytes / 4;
        const hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) { pinned = nullptr; return e; }
        pinned_cap = want;
        return hipSuccess;
    }
    // stream2 (a pairing batch's G1 membership kernel, the copies of a staged call) and stream3 (the side chain of the two-level
    // bucket reduce: highest priority the device offers, so that its waves are placed WHILE the main stream's accumulate still has
    // blocks to dispatch) are created by the first pipeline that needs them.  Round 4: a slot that only ever serves small calls
    // keeps ONE stream -- with three per slot, one of them a priority stream, 64 native callers of 128-record multiexps ran at
    // 1.1-1.2e4 calls/s against 2.0-2.5e4 before the priority streams existed (profiles/r04_concurrent_callers_native.txt).
    hipError_t need_stream2() {
        if (stream2) return hipSuccess;
        return hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking);
    }
    hipError_t need_stream3() {
        if (stream3) return hipSuccess;
        int lo = 0, hi = 0;                              // numerically lowest = highest priority
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); hi = 0; }
        if (hipStreamCreateWithPriority(&stream3, hipStreamNonBlocking, hi) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking);
    }

    template <class Fn> void for_each_buf(Fn &&fn) {
        for (DevBuf *b : {&input, &misc, &pts, &digits, &hist16, &slice_base, &counts, &offsets, &taskoff, &scan_blk,
                          &entries, &tasks, &perm, &split_lists, &partial, &winout, &bacc, &taskbkt, &rcsum}) fn(*b);
    }
    size_t workspace_bytes() { size_t t = 0; for_each_buf([&](DevBuf &b) { t += b.cap; }); return t; }
    void release_workspace() { for_each_buf([](DevBuf &b) { b.release(); }); }
};

// Async device-to-host copies target host objects owned by the calling frame: if a later HIP call fails and the function
// returns early, the stream is drained first so that no copy is still writing into a dead stack / vector.
struct StreamDrain {
    hipStream_t s;
    bool armed = true;
    ~StreamDrain() { if (armed) (void)hipStreamSynchronize(s); }
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
    uint32_t slice;    // records per block of the counting-sort kernels (msm.hip: msm_slice_for)
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

// Coalesced batch of M small MSMs (every call below 2048 records; msm.hip "coalesced batch"): d_in holds
// the calls' records back to back, coff[0..M] their record offsets (host).  Writes rc[j] and, per call,
// kMsmBatchWindows XYZZ window sums (8-bit windows, lowest first) to wins_words; the caller applies
// Horner.  Non-zero return: HIP failure, every call of the batch fails.
static constexpr int kMsmBatchWindows = 32, kMsmBatchWindowBits = 8, kMsmBatchMaxCalls = 64;
int msm_g1_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *wins_words, int *rc);
int msm_g2_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *wins_words, int *rc);

// Pairing: on success returns 0 and writes the product of the Miller loops (before the final
// exponentiation) as 144 Montgomery words to `ml_words`.
int pairing_device(Engine *e, const void *d_in, size_t k, uint32_t *ml_words);

// Coalesced batch of M small pairing calls (pairing.hip): d_in holds the calls' pairs back to back, coff[0..M]
// their pair offsets (host).  Writes rc[j] and per call the 68 per-step line products (Fp12, 144 words each)
// to L_words[(j * 68 + step) * 144]; the caller applies miller_product_from_steps and the final exponentiation.
static constexpr int kPairBatchMaxCalls = 64, kPairBatchMaxPairs = 64, kPairSteps = 68;
int pairing_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *L_words, int *rc);

}  // namespace eip
