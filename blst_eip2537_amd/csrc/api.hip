// C-ABI of libeip2537_hip.so: the reference's 13 precompile entry points (include/eip2537.h),
// its gas API, and the device / sharding extensions (include/eip2537_hip.h).
//
// Length rules, validation order and error codes restate the reference's precompile logic:
//   bls12_g1add / g1mul        reference src/eip2537.c:434-524
//   bls12_g1multiexp*          reference src/eip2537.c:541-708  (here: one GPU Pippenger for all n)
//   bls12_g2add / g2mul / g2multiexp*   reference src/eip2537.c:722-998
//   bls12_pairing              reference src/eip2537.c:1020-1081
//   bls12_map_*                reference src/eip2537.c:1094-1165  (host: csrc/h2c.h)
//   gas                        reference src/eip2537.c:1168-1271
// Single-pair operations (add, mul, map) are host code of this library (BASELINE config 1 is
// "CPU plumbing, no GPU"); the multiexp and pairing paths have no CPU implementation here at all:
// without a working HIP device they fail loudly with EIP2537_MEMORY_ERROR.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <type_traits>
#include <vector>
#include "codec.h"
#include "pairing.h"
#include "ifma.h"
#include "ifma_horner.h"
#include "h2c.h"
#include "limbk.h"
#include "engine.h"
#include "../../include/eip2537.h"
#include "../../include/eip2537_hip.h"

#define API extern "C" __attribute__((visibility("default")))

namespace eip {

hipError_t DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    fresh = true;
    return hipSuccess;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// Engine pools.  The ABI has no handle to hang state on and its callers are concurrent (cargo's
// test threads, goroutines: SURVEY.md 8b "Threading"), so the library keeps, per HIP device, a
// small pool of engines -- each with its own streams, events and grow-only workspace -- and a call
// borrows one for its duration.  Calls on different slots overlap on the GPU (small inputs fill a
// fraction of the 256 CUs); when every slot of a pool is busy a caller waits for the next release.
//
// Devices.  `g_split` lists the pools a HOST-input call may use: $EIP2537_HIP_DEVICES ("all" or a
// comma list of ordinals; an ordinal may repeat, which gives it a second pool -- used by the tests to
// run the multi-device split on one GPU), else the single device of eip2537_hip_init(d) /
// $EIP2537_HIP_DEVICE, else every visible device.  Large host inputs are cut into contiguous record
// ranges, one per listed device and host thread (each device copies its range over its own PCIe
// link); small ones go to the least busy pool.  Device-input calls (`*_dev`) run on the device that
// owns the pointer.
static constexpr int kMaxSlots = 16;
static constexpr int kMaxPools = 32;
struct DevicePool {
    int ordinal = -1;
    Engine slots[kMaxSlots];
    bool busy[kMaxSlots] = {};
    int nbusy = 0;
};
static std::mutex g_mu;                 // guards the pool tables and one-time device selection
static std::condition_variable g_cv;
static DevicePool g_pools[kMaxPools];
static int g_npools = 0;
static int g_split[kMaxPools];          // pool indices host-input calls are spread over
static int g_nsplit = 0;                // 0 = devices not selected yet
static int g_nslots = 8;
static int g_ndev = 0;
static int g_device_request = -1;
static size_t g_keep_bytes = (size_t)4096 << 20;     // per-slot workspace kept between calls
static std::atomic<int> g_window_override{0};
static std::atomic<int> g_route_override{-1};        // -1 default, 0 always GPU, 1 always host (small-call tests)

struct CallStats {
    float pipeline_ms = 0.f, dominant_ms = 0.f, aux_ms[2] = {0.f, 0.f};
    LastPlan plan{};
    bool valid = false;
};
static thread_local CallStats t_last;
static CallStats g_last;

// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told
// otherwise) and streams sharing a queue run one after the other; every slot drives up to three
// streams.  The variable is read when the runtime initialises, so it is set (never overriding the
// embedder's own value) when this library is loaded -- before its first HIP call, and for a linked
// binary before main() starts any thread -- instead of from inside a precompile call.
// (Round 3: the library no longer calls setenv itself -- through the static shim it is dlopen'ed lazily from caller
// threads, where setenv races with getenv in the embedder's runtime.  The embedder sets GPU_MAX_HW_QUEUES=16 before its
// first HIP call; blst_eip2537_amd/__init__.py, bench.py and the tests do; the first engine warns once
// when it is unset.)
static void warn_hw_queues_once() {
    static std::atomic<bool> said{false};
    if (!getenv("GPU_MAX_HW_QUEUES") && !said.exchange(true))
        fprintf(stderr, "[eip2537_hip] note: GPU_MAX_HW_QUEUES is not set; concurrent callers share 4 hardware queues "
                        "(set GPU_MAX_HW_QUEUES=16 before the first HIP call, see INTEGRATION.md)\n");
}

// Make `dev` current on this thread for the duration of a call and put the caller's device back
// afterwards (an embedder such as torch keeps its own notion of the current device).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

static int pool_new_locked(int ordinal) {
    if (g_npools >= kMaxPools) return -1;
    g_pools[g_npools].ordinal = ordinal;
    return g_npools++;
}
static int pool_for_ordinal_locked(int ordinal) {
    for (int i = 0; i < g_npools; i++)
        if (g_pools[i].ordinal == ordinal) return i;
    return pool_new_locked(ordinal);
}
static bool device_select_locked() {
    if (g_nsplit) return true;
    int ndev = 0;
    hipError_t er = hipGetDeviceCount(&ndev);
    if (er != hipSuccess || ndev == 0) {
        fprintf(stderr, "[eip2537_hip] FATAL: no HIP device available (%s); the multiexp / pairing "
                        "path has no CPU fallback\n", er == hipSuccess ? "0 devices" : hipGetErrorString(er));
        return false;
    }
    g_ndev = ndev;
    int want[kMaxPools], nwant = 0;
    const char *list = getenv("EIP2537_HIP_DEVICES");
    if (list && !*list) list = nullptr;                       // an empty value means "not set"
    if (g_device_request >= 0) {
        want[nwant++] = g_device_request;
    } else if (list && *list && strcmp(list, "all") != 0) {
        for (const char *c = list; *c && nwant < kMaxPools;) {
            char *endp = nullptr;
            long v = strtol(c, &endp, 10);
            if (endp == c) { fprintf(stderr, "[eip2537_hip] FATAL: cannot parse EIP2537_HIP_DEVICES=\"%s\"\n", list); return false; }
            want[nwant++] = (int)v;
            c = *endp == ',' ? endp + 1 : endp;
            if (*endp && *endp != ',') { fprintf(stderr, "[eip2537_hip] FATAL: cannot parse EIP2537_HIP_DEVICES=\"%s\"\n", list); return false; }
        }
    } else if (const char *one = (list ? nullptr : getenv("EIP2537_HIP_DEVICE"))) {
        want[nwant++] = atoi(one);
    } else {
        for (int d = 0; d < ndev && nwant < kMaxPools; d++) want[nwant++] = d;
    }
    for (int i = 0; i < nwant; i++)
        if (want[i] < 0 || want[i] >= ndev) {
            fprintf(stderr, "[eip2537_hip] FATAL: HIP device ordinal %d requested, %d device(s) visible\n", want[i], ndev);
            return false;
        }
    const char *env = getenv("EIP2537_HIP_SLOTS");
    int ns = env ? atoi(env) : 8;
    g_nslots = ns < 1 ? 1 : ns > kMaxSlots ? kMaxSlots : ns;
    if ((env = getenv("EIP2537_HIP_KEEP_MB"))) g_keep_bytes = (size_t)strtoull(env, nullptr, 10) << 20;
    int ns_out = 0;
    for (int i = 0; i < nwant; i++) {
        bool repeat = false;
        for (int j = 0; j < i; j++) repeat |= want[j] == want[i];
        int pi = repeat ? pool_new_locked(want[i]) : pool_for_ordinal_locked(want[i]);
        if (pi < 0) return false;
        g_split[ns_out++] = pi;
    }
    g_nsplit = ns_out;
    return g_nsplit > 0;
}
static bool slot_init(Engine &e, int ordinal) {
    if (e.ready) return true;
    warn_hw_queues_once();
    e.device = ordinal;
    // (stream2 / stream3 are created by the first pipeline that needs them: Engine::need_stream2 / need_stream3)
    bool ok = hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&e.ev_start) == hipSuccess && hipEventCreate(&e.ev_stop) == hipSuccess &&
              hipEventCreate(&e.ev_a) == hipSuccess && hipEventCreate(&e.ev_b) == hipSuccess &&
              hipEventCreate(&e.ev_j2) == hipSuccess && hipEventCreate(&e.ev_j3) == hipSuccess && hipEventCreate(&e.ev_c) == hipSuccess;
    if (!ok) { fprintf(stderr, "[eip2537_hip] FATAL: stream/event creation failed on device %d\n", ordinal); return false; }
    e.ready = true;
    e.failed = false;
    return true;
}
// After a HIP failure in the middle of a pipeline: wait for whatever is still in flight on any of
// the slot's streams (a forked kernel may still be writing the error word), then drop the streams,
// events and workspace so that the next borrower starts from a freshly created engine.
static void slot_reset(Engine &e) {
    if (e.stream) (void)hipStreamSynchronize(e.stream);
    if (e.stream2) (void)hipStreamSynchronize(e.stream2);
    if (e.stream3) (void)hipStreamSynchronize(e.stream3);
    (void)hipGetLastError();
    for (hipStream_t *s : {&e.stream, &e.stream2, &e.stream3}) { if (*s) (void)hipStreamDestroy(*s); *s = nullptr; }
    for (hipEvent_t *v : {&e.ev_start, &e.ev_stop, &e.ev_a, &e.ev_b, &e.ev_j2, &e.ev_j3, &e.ev_c}) { if (*v) (void)hipEventDestroy(*v); *v = nullptr; }
    for (hipEvent_t &v : e.ev_copy) { if (v) (void)hipEventDestroy(v); v = nullptr; }
    if (e.pinned) { (void)hipHostFree(e.pinned); e.pinned = nullptr; e.pinned_cap = 0; }
    for (hipEvent_t &v : e.ev_sorted) { if (v) (void)hipEventDestroy(v); v = nullptr; }
    for (hipEvent_t &v : e.ev_accdone) { if (v) (void)hipEventDestroy(v); v = nullptr; }
    e.release_workspace();
    e.ready = false;
    e.failed = false;
}
// Borrow an engine of pool `pi` for one call (RAII).  `e` is null when no device is usable.
struct SlotLease {
    Engine *e = nullptr;
    int pi = -1, idx = -1;
    DeviceGuard *guard = nullptr;
    explicit SlotLease(int pool_index) {
        std::unique_lock<std::mutex> lk(g_mu);
        if (!device_select_locked()) return;
        if (pool_index < 0) {                      // any listed device: the least busy pool
            pool_index = g_split[0];
            for (int i = 1; i < g_nsplit; i++)
                if (g_pools[g_split[i]].nbusy < g_pools[pool_index].nbusy) pool_index = g_split[i];
        }
        pi = pool_index;
        DevicePool &p = g_pools[pi];
        for (;;) {
            for (int i = 0; i < g_nslots; i++)
                if (!p.busy[i]) { idx = i; break; }
            if (idx >= 0) break;
            g_cv.wait(lk);
        }
        p.busy[idx] = true;
        p.nbusy++;
        lk.unlock();
        // the slot is ours now: create its streams outside the table lock
        guard = new DeviceGuard(p.ordinal);
        if (guard->ok && slot_init(p.slots[idx], p.ordinal)) { e = &p.slots[idx]; return; }
        delete guard;
        guard = nullptr;
        lk.lock();
        p.busy[idx] = false;
        p.nbusy--;
        idx = -1;
        g_cv.notify_all();
    }
    ~SlotLease() {
        if (idx < 0) return;
        t_last.pipeline_ms = e->last_kernel_ms;
        t_last.dominant_ms = e->last_accum_ms;
        t_last.aux_ms[0] = e->last_aux_ms[0];
        t_last.aux_ms[1] = e->last_aux_ms[1];
        t_last.plan = e->last_plan;
        t_last.valid = true;
        if (e->failed) slot_reset(*e);
        else if (e->workspace_bytes() > g_keep_bytes) e->release_workspace();
        delete guard;
        std::lock_guard<std::mutex> lk(g_mu);
        g_last = t_last;
        g_pools[pi].busy[idx] = false;
        g_pools[pi].nbusy--;
        g_cv.notify_all();
    }
    SlotLease(const SlotLease &) = delete;
    SlotLease &operator=(const SlotLease &) = delete;
};
// pool of the device that owns a device pointer (for the *_dev entry points); -1 = not device memory
static int pool_of_pointer(const void *d_ptr) {
    hipPointerAttribute_t at;
    int ordinal = -1;
    if (hipPointerGetAttributes(&at, d_ptr) == hipSuccess && at.type == hipMemoryTypeDevice) ordinal = at.device;
    else (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_mu);
    if (!device_select_locked()) return -1;
    if (ordinal < 0) return -2;
    // a single-device process (one rank per GPU) keeps using its own pool
    for (int i = 0; i < g_nsplit; i++)
        if (g_pools[g_split[i]].ordinal == ordinal) return g_split[i];
    return pool_for_ordinal_locked(ordinal);
}

// Host -> device staging of the caller's buffer on the engine's own stream: the kernels that read it
// are ordered behind the copy by the stream, and the call does not return before its last kernel has
// finished, so the caller's pointer is never retained (Go / Rust own the memory: SURVEY.md 8b).
static int stage_input(Engine *e, const void *in, size_t len) {
    if (e->input.reserve(len) != hipSuccess) {
        fprintf(stderr, "[eip2537_hip] hipMalloc of %zu staging bytes failed\n", len);
        e->failed = true;
        return E_MEMORY_ERROR;
    }
    if (hipMemcpyAsync(e->input.p, in, len, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
        fprintf(stderr, "[eip2537_hip] host-to-device copy failed\n");
        e->failed = true;
        return E_MEMORY_ERROR;
    }
    return E_SUCCESS;
}

// ---- host-side single operations -------------------------------------------------------
template <class F> static int host_decode_point(Aff<F> &out, const byte *in) {
    uint32_t w[Wire<F>::kPointWords];
    memcpy(w, in, sizeof w);
    return decode_point<F>(out, w);
}
template <class F> static void host_encode_point(byte *out, const Aff<F> &a) {
    uint32_t w[Wire<F>::kPointWords];
    encode_point<F>(w, a);
    memcpy(out, w, sizeof w);
}
template <class F> static int host_add(byte *out, const byte *in, size_t in_len) {
    const size_t pb = Wire<F>::kPointWords * 4;
    if (in_len != 2 * pb) return E_INVALID_LENGTH;
    Aff<F> a, b;
    int st = host_decode_point<F>(a, in);
    if (st) return st;
    st = host_decode_point<F>(b, in + pb);
    if (st) return st;
    host_encode_point<F>(out, to_affine(madd(from_affine(a), b)));
    return E_SUCCESS;
}
template <class F> static int host_mul(byte *out, const byte *in, size_t in_len) {
    const size_t pb = Wire<F>::kPointWords * 4;
    if (in_len != pb + 32) return E_INVALID_LENGTH;
    Aff<F> a;
    int st = host_decode_point<F>(a, in);
    if (st) return st;
    uint32_t sw[8], k[8];
    memcpy(sw, in + pb, 32);
    decode_scalar(k, sw);
    // G2: the windowed form (260 doublings in chains of 5 on IFMA vectors + 52 additions) instead of 256 doublings + ~128 mixed additions
    if (std::is_same<F, Fp2>::value) host_encode_point<F>(out, to_affine(msm_interleaved<F>(&a, k, 1)));
    else host_encode_point<F>(out, to_affine(scalar_mul(a, k, 256)));
    return E_SUCCESS;
}

template <class F> static int msm_dispatch(Engine *e, const void *d_in, size_t n, uint32_t *pw);
template <> int msm_dispatch<Fp>(Engine *e, const void *d_in, size_t n, uint32_t *pw) { return msm_g1_device(e, d_in, n, pw, g_window_override.load()); }
template <> int msm_dispatch<Fp2>(Engine *e, const void *d_in, size_t n, uint32_t *pw) { return msm_g2_device(e, d_in, n, pw, g_window_override.load()); }

// One device pipeline on pool `pi` (-1: least busy listed device).  device_input: `in` is already in
// HBM on that pool's device; want_partial: write the projective partial instead of the encoding.
template <class F>
static int msm_entry(int pi, byte *out, const void *in, size_t n, bool device_input, bool want_partial, CopyGate *gate = nullptr, int turn = 0,
                     const ShardFeed *feed = nullptr) {
    // a shard of a pipelined call takes its engine slot only when it is its turn to copy: a slot holder never waits for another
    // shard, so concurrent pipelined calls cannot starve each other of slots
    if (gate) gate->wait_turn(turn);
    SlotLease lease(pi);
    Engine *e = lease.e;
    if (!e) return E_MEMORY_ERROR;
    const void *d_in = in;
    if (!device_input) {
        // the device pipeline stages the buffer itself, chunk by chunk, with the decode of a chunk behind its copy (msm.hip)
        if (e->input.reserve(n * Wire<F>::kMsmRecWords * 4) != hipSuccess) {
            fprintf(stderr, "[eip2537_hip] hipMalloc of %zu staging bytes failed\n", n * (size_t)Wire<F>::kMsmRecWords * 4);
            e->failed = true;
            return E_MEMORY_ERROR;
        }
        e->host_src = in;
        e->copy_gate = gate;
        e->copy_turn = turn;
        d_in = e->input.p;
    }
    if (feed) e->feed = *feed;                      // record shards of one bucket space (msm.hip): staged copies | sort beside accumulate
    Xyzz<F> acc;
    int st = msm_dispatch<F>(e, d_in, n, reinterpret_cast<uint32_t *>(&acc));
    e->host_src = nullptr;                      // never retained past the call (an early error return leaves it set)
    e->copy_gate = nullptr;
    e->feed.k = 0;
    if (st) return st;
    if (want_partial) {
        memcpy(out, &acc, sizeof acc);
    } else {
        host_encode_point<F>(out, to_affine(acc));
    }
    return E_SUCCESS;
}
static std::vector<uint32_t> parse_weights(const char *v) {
    std::vector<uint32_t> w;
    if (v)
        for (const char *c = v; *c;) {
            char *endp = nullptr;
            const unsigned long x = strtoul(c, &endp, 10);
            if (endp == c) break;
            w.push_back((uint32_t)x);
            c = *endp == ',' ? endp + 1 : endp;
        }
    return w;
}
static ShardFeed feed_from_weights(size_t n, const std::vector<uint64_t> &w) {
    ShardFeed f;
    uint64_t total = 0;
    for (uint64_t x : w) total += x;
    if (w.size() < 2 || !total) return f;
    uint64_t run = 0;
    f.bound[0] = 0;
    int k = 0;
    for (size_t i = 0; i < w.size(); i++) {
        run += w[i];
        const uint32_t b = i + 1 == w.size() ? (uint32_t)n : (uint32_t)((unsigned __int128)n * run / total);
        if (b > f.bound[k]) f.bound[++k] = b;          // empty shards vanish
    }
    f.k = k;
    return f;
}
// Record shards of a device-resident G1 call (EIP2537_DEV_STAGES=k | w0,w1,...; 0 or 1: one shard).  Default: from 2^21 records equal
// shards of 2^19 (at most 8).  Measured (profiles/r04_dev_shards.txt): 2^21 6.06 -> 5.88 ms, 2^22 11.9 -> 10.9 ms; at 2^20 every cut
// loses (3.22 -> 3.29 .. 3.6: a shard's extra tasks and bucket-accumulator round trips cost more than the hidden sort stage saves).
static ShardFeed stage_plan_dev(size_t n) {
    static const std::vector<uint32_t> env = parse_weights(getenv("EIP2537_DEV_STAGES"));
    if (g_window_override.load() != 0 || n >= ((size_t)1 << 31)) return ShardFeed{};
    std::vector<uint64_t> w;
    if (env.size() == 1) { if (env[0] >= 2) w.assign(std::min<size_t>(env[0], ShardFeed::kMax), 1u); }
    else if (env.size() > 1) w.assign(env.begin(), env.begin() + std::min<size_t>(env.size(), ShardFeed::kMax));
    else if (n >= ((size_t)1 << 21)) w.assign(std::min<size_t>(ShardFeed::kMax, n >> 19), 1u);
    return feed_from_weights(n, w);
}
template <class F> static int msm_dev_abi(byte *out, const void *d_in, size_t n, bool want_partial) {
    if (!n) return E_INVALID_LENGTH;
    int pi = pool_of_pointer(d_in);
    if (pi == -2) { fprintf(stderr, "[eip2537_hip] %p is not device memory\n", d_in); return E_MEMORY_ERROR; }
    if (pi < 0) return E_MEMORY_ERROR;
    if (std::is_same<F, Fp>::value) {
        // a large device-resident G1 input is cut into record shards of one bucket space too: the sort stage of shard s + 1 then runs
        // beside the accumulate of shard s (msm.hip), and every shard's limb records stay inside the last-level cache
        const ShardFeed feed = stage_plan_dev(n);
        if (feed.k > 1) return msm_entry<F>(pi, out, d_in, n, true, want_partial, nullptr, 0, &feed);
    }
    return msm_entry<F>(pi, out, d_in, n, true, want_partial);
}

// ---- crossover: calls too small for a launch ------------------------------------
// The EVM sends mostly tiny inputs (reference README.md:33; bench sizes from 2 pairs up,
// rust/benches/eip2537_benches.rs:69-70,134,177).  A GPU call costs its fixed chain latency whatever
// its size (G1 MSM ~0.37 ms, G2 MSM ~0.9 ms, pairing ~1.9 ms: profiles/r02_small_calls.txt), so below
// the measured crossover the library runs its own host code (curve.h / pairing.h -- the same code the
// single-pair precompiles use, never anything under oracle/): one interleaved windowed multiplication
// over the records, and for a pairing the reference's own sequence (src/eip2537.c:1033-1070).  The
// reference makes the same kind of cut (n == 1 forwards to the mul precompile, :550-552).
// This is a size rule inside a working engine, not a fallback: without a usable HIP device these
// calls still fail loudly (the device is selected first), and every size above the crossover has no
// host path at all.  eip2537_hip_set_route() pins the route for tests.
// round 3 (profiles/r03_small_calls.txt): the pairing pipeline's fixed latency fell from 1.9 to 0.8 ms, so the host route now
// ends at 2 pairs (0.67 ms; 3 pairs: 0.91 on the host against 0.85), and at 6 G2 records (8: 0.67 against 0.63)
static constexpr size_t kHostMaxG1 = 16, kHostMaxG2 = 6, kHostMaxPairs = 2;
static constexpr size_t kHostRouteTestMax = 64;        // route 1 ("host whenever allowed") still refuses more
template <class F> struct HostMax { static constexpr size_t kUnits = kHostMaxG1; };
template <> struct HostMax<Fp2> { static constexpr size_t kUnits = kHostMaxG2; };
static bool host_route(size_t units, size_t crossover) {
    const int r = g_route_override.load();
    if (r == 0) return false;
    if (r == 1) return units <= kHostRouteTestMax;
    return units <= crossover;
}
static bool device_present() {
    std::lock_guard<std::mutex> lk(g_mu);
    return device_select_locked();
}
// Host MSM of the small-call route: msm_interleaved() of curve.h (interleaved signed 5-bit windows, Straus:
// one chain of 260 doublings shared by all records, per record a table of 1..16 multiples and at most 52
// additions) -- 57 + 22 n microseconds for G1 on the GPU box's host against 100 n for one double-and-add
// per record, which moves the crossover with the GPU's fixed ~0.4 ms (G2 ~0.75 ms) from 2 records to 16 (G2: 8).
template <class F> static int msm_host_small(byte *out, const byte *in, size_t n) {
    const size_t rec = Wire<F>::kMsmRecWords * 4, pb = Wire<F>::kPointWords * 4;
    std::vector<Aff<F>> pts(n);
    std::vector<uint32_t> ks(n * 8);
    for (size_t i = 0; i < n; i++, in += rec) {
        int st = host_decode_point<F>(pts[i], in);
        if (st) return st;                                  // first bad record in input order
        uint32_t sw[8];
        memcpy(sw, in + pb, 32);
        decode_scalar(&ks[i * 8], sw);
    }
    host_encode_point<F>(out, to_affine(msm_interleaved<F>(pts.data(), ks.data(), n)));
    return E_SUCCESS;
}
static void pairing_finish(byte *out, const Fp12 &ml);
static int pairing_host_small(byte *out, const byte *in, size_t k) {
    if (k > kHostRouteTestMax) return E_MEMORY_ERROR;
    Aff<Fp> P[kHostRouteTestMax];
    Aff<Fp2> Q[kHostRouteTestMax];
    for (size_t i = 0; i < k; i++, in += 384) {              // the reference's order of checks, pair by pair (src/eip2537.c:1033-1053)
        int st = host_decode_point<Fp>(P[i], in);
        if (st) return st;
        if (!in_g1(P[i])) return E_NOT_IN_SUBGROUP;
        st = host_decode_point<Fp2>(Q[i], in + 128);
        if (st) return st;
        if (!in_g2_host(Q[i])) return E_NOT_IN_SUBGROUP;
    }
    pairing_finish(out, miller_loop_multi(P, Q, k));          // one chain of squarings for all pairs
    return E_SUCCESS;
}

// ---- coalescing of concurrent small MSM calls -------------------------------------
// A small GPU call is all fixed latency (~0.4-0.55 ms for anything up to a few hundred records), and the
// real callers are concurrent (goroutines / test threads under an EVM).  So small host-input MSMs do
// not each take an engine: a caller queues its request, and whichever waiting caller finds a free
// "flight" becomes the leader of everything queued at that moment -- one staging copy, ONE device
// pipeline over the concatenated records with per-call bucket ranges and per-call error words
// (msm.hip, "coalesced batch") -- then hands every caller its 32 window sums; each caller finishes its
// own Horner, inversion and encoding on its own thread.  Batches form by themselves: while the flights
// are busy, arrivals pile up for the next leader; an uncontended caller leads a batch of one, which
// takes the ordinary single-call path.  $EIP2537_HIP_COALESCE=0 turns the queue off.
template <class F> struct CoalesceCfg { static constexpr size_t kMinRecords = 2, kMaxRecords = 512; };
// Device pipelines allowed in flight for these calls, by how many callers are outstanding (queued or being
// served): few callers each get a pipeline at once (what the engine slots gave them before); the more
// callers pile up, the fewer and larger the batches (measured, 128-record G1 calls: 64 callers 28 k
// calls/s with 2 flights, 20 k with 4, 7 k uncoalesced; 4 callers 6.6 k with >= 4 flights, 3.8 k with 2).
static int coalesce_flights(size_t outstanding) { return outstanding <= 8 ? 8 : outstanding <= 24 ? 4 : 2; }
static std::atomic<uint64_t> g_co_batches{0}, g_co_calls{0}, g_co_max{0};   // device pipelines run / calls served / largest batch
static constexpr size_t kCoalesceMaxRecords = 16384;          // per batch
template <class F> struct MsmReq {
    const byte *in = nullptr;
    size_t n = 0;
    Xyzz<F> wins[kMsmBatchWindows];
    Xyzz<F> whole;                   // batch of one: the ordinary pipeline's partial sum
    bool have_whole = false;
    int rc = E_MEMORY_ERROR;
    bool taken = false, done = false;
};
template <class Req> struct Batcher {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Req *> pending;
    int in_flight = 0;           // device pipelines running
    size_t served = 0;           // calls inside those pipelines
};
template <class Req> static Batcher<Req> &batcher() { static Batcher<Req> b; return b; }
static bool coalesce_enabled() {
    static const bool on = [] { const char *v = getenv("EIP2537_HIP_COALESCE"); return !v || atoi(v) != 0; }();
    return on;
}
template <class F> static int msm_batch_dispatch(Engine *e, const void *d_in, const uint32_t *coff, int M, Xyzz<F> *wins, int *rc);
template <> int msm_batch_dispatch<Fp>(Engine *e, const void *d_in, const uint32_t *coff, int M, Xyzz<Fp> *wins, int *rc) {
    return msm_g1_batch_device(e, d_in, coff, M, reinterpret_cast<uint32_t *>(wins), rc);
}
template <> int msm_batch_dispatch<Fp2>(Engine *e, const void *d_in, const uint32_t *coff, int M, Xyzz<Fp2> *wins, int *rc) {
    return msm_g2_batch_device(e, d_in, coff, M, reinterpret_cast<uint32_t *>(wins), rc);
}
template <class F> static void run_msm_batch(std::vector<MsmReq<F> *> &batch) {
    const size_t rec = Wire<F>::kMsmRecWords * 4;
    const int M = (int)batch.size();
    SlotLease lease(-1);
    Engine *e = lease.e;
    if (!e) { for (auto *r : batch) r->rc = E_MEMORY_ERROR; return; }
    if (M == 1) {                       // nobody to share with: the ordinary pipeline
        MsmReq<F> *r = batch[0];
        int st = stage_input(e, r->in, r->n * rec);
        if (!st) st = msm_dispatch<F>(e, e->input.p, r->n, reinterpret_cast<uint32_t *>(&r->whole));
        r->have_whole = st == E_SUCCESS;
        r->rc = st;
        return;
    }
    std::vector<uint32_t> coff((size_t)M + 1, 0u);
    for (int j = 0; j < M; j++) coff[(size_t)j + 1] = coff[(size_t)j] + (uint32_t)batch[(size_t)j]->n;
    const size_t total = coff[(size_t)M];
    int st = e->input.reserve(total * rec) == hipSuccess ? E_SUCCESS : E_MEMORY_ERROR;
    for (int j = 0; j < M && !st; j++)      // one H2D per call, back to back on the engine's stream
        if (hipMemcpyAsync(static_cast<char *>(e->input.p) + (size_t)coff[(size_t)j] * rec, batch[(size_t)j]->in,
                           batch[(size_t)j]->n * rec, hipMemcpyHostToDevice, e->stream) != hipSuccess) st = E_MEMORY_ERROR;
    std::vector<Xyzz<F>> wins((size_t)M * kMsmBatchWindows);
    std::vector<int> rc((size_t)M, E_MEMORY_ERROR);
    if (st) e->failed = true;
    else st = msm_batch_dispatch<F>(e, e->input.p, coff.data(), M, wins.data(), rc.data());
    for (int j = 0; j < M; j++) {
        MsmReq<F> *r = batch[(size_t)j];
        r->rc = st ? st : rc[(size_t)j];
        if (!r->rc) memcpy(r->wins, &wins[(size_t)j * kMsmBatchWindows], sizeof r->wins);
    }
}
// Queue `req`, lead a batch when a flight is free, return when some leader (possibly this thread) has
// served it.  `run(batch)` executes one device pipeline for the batch and fills every request's result.
template <class Req, class Run>
static void coalesce(Req &req, size_t max_calls, size_t max_units, Run &&run) {
    Batcher<Req> &b = batcher<Req>();
    std::unique_lock<std::mutex> lk(b.mu);
    b.pending.push_back(&req);
    while (!req.done) {
        if (!req.taken && b.in_flight < coalesce_flights(b.served + b.pending.size())) {
            // lead: this request first, then the queue in arrival order, up to the batch limits
            std::vector<Req *> batch{&req};
            size_t total = req.n;
            req.taken = true;
            std::vector<Req *> rest;
            for (Req *r : b.pending) {
                if (r == &req) continue;
                if (batch.size() < max_calls && total + r->n <= max_units) { r->taken = true; batch.push_back(r); total += r->n; }
                else rest.push_back(r);
            }
            b.pending.swap(rest);
            b.in_flight++;
            b.served += batch.size();
            // requests the batch limits left behind can be led by one of their own callers as long as a flight is free
            if (!b.pending.empty() && b.in_flight < coalesce_flights(b.served + b.pending.size())) b.cv.notify_all();
            lk.unlock();
            g_co_batches++;
            g_co_calls += batch.size();
            for (uint64_t m = g_co_max.load(); batch.size() > m && !g_co_max.compare_exchange_weak(m, batch.size());) {}
            try {
                run(batch);
            } catch (...) {                 // std::bad_alloc in a leader must not strand its followers or cross the C ABI
                for (Req *r : batch) r->rc = E_MEMORY_ERROR;
            }
            lk.lock();
            b.in_flight--;
            b.served -= batch.size();
            for (Req *r : batch) r->done = true;
            b.cv.notify_all();
        } else {
            b.cv.wait(lk);
        }
    }
}
template <class F> static int msm_coalesced(byte *out, const byte *in, size_t n) {
    MsmReq<F> req;
    req.in = in;
    req.n = n;
    coalesce(req, (size_t)kMsmBatchMaxCalls, kCoalesceMaxRecords, [](std::vector<MsmReq<F> *> &batch) { run_msm_batch<F>(batch); });
    if (req.rc) return req.rc;
    Xyzz<F> acc;
    if (req.have_whole) {
        acc = req.whole;
    } else {                            // Horner over the 32 window sums, highest window first
        HornerAcc<F> h;
        for (int w = kMsmBatchWindows - 1; w >= 0; w--) {
            h.dbl_n(kMsmBatchWindowBits);
            h.add(req.wins[w]);
        }
        acc = h.result();
    }
    host_encode_point<F>(out, to_affine(acc));
    return E_SUCCESS;
}

// ---- record-range split over the listed devices -------------------------------------
// A host-input call with at least two shards' worth of records is cut into contiguous ranges, one per
// listed device; each range is staged and reduced by its own host thread on its own device (its own
// PCIe link), down to one projective partial (MSM) or one Fp12 Miller product (pairing); the caller's
// thread runs shard 0 and combines.  Point addition / Fp12 multiplication of N partials on the host is
// what replaces a collective inside one process (no RCCL: there is no second process to talk to).
// Errors: shards are ordered record ranges, so the lowest-index bad record of the whole input is the
// one reported by the first failing shard.
// Minimum records per shard, from the single-GPU size sweep (profiles/r01_size_sweep.txt: below these
// sizes a call costs its fixed chain latency whatever its size, so cutting it further buys nothing).
template <class F> struct SplitMin { static constexpr size_t kRecords = (size_t)1 << 16; };
template <> struct SplitMin<Fp2> { static constexpr size_t kRecords = (size_t)1 << 15; };
static constexpr size_t kPairingSplitMin = 1024;
static size_t split_min_override() {
    static const size_t v = [] { const char *e = getenv("EIP2537_HIP_SPLIT_MIN"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)0; }();
    return v;
}
// pools to use for n units: empty => one pipeline on the least busy device
static std::vector<int> split_plan(size_t n, size_t min_per_shard) {
    std::vector<int> pools;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!device_select_locked()) return pools;
        if (split_min_override()) min_per_shard = split_min_override();
        size_t shards = std::min<size_t>((size_t)g_nsplit, n / std::max<size_t>(1, min_per_shard));
        if (shards < 2) return pools;
        pools.assign(g_split, g_split + shards);
    }
    return pools;
}
// Persistent worker threads for the shards of split calls (round 4: round 3 created and joined std::threads inside every such
// call).  A job never waits in the queue: post() starts another worker whenever no idle one is left (up to kMaxWorkers; beyond
// that it refuses and the caller runs the shard itself).  Leaked on purpose: workers sleep on its condition variable until the
// process ends.
class Workers {
    static constexpr int kMaxWorkers = 64;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    int idle = 0, total = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            idle++;
            cv.wait(lk, [&] { return !q.empty(); });
            idle--;
            std::function<void()> f = std::move(q.front());
            q.pop_front();
            lk.unlock();
            f();
            lk.lock();
        }
    }
public:
    bool post(std::function<void()> f) {
        std::lock_guard<std::mutex> lk(m);
        if ((int)q.size() >= idle) {                 // every idle worker already has a job coming
            if (total >= kMaxWorkers) return false;
            try { std::thread([this] { loop(); }).detach(); } catch (...) { return false; }
            total++;
        }
        q.push_back(std::move(f));
        cv.notify_one();
        return true;
    }
    static Workers &get() { static Workers *w = new Workers; return *w; }
};
// run fn(shard) for every shard, shard 0 on the calling thread, the others on the persistent workers
template <class Fn> static void run_shards(size_t shards, Fn &&fn) {
    std::vector<CallStats> stats(shards);
    std::mutex dm;
    std::condition_variable dcv;
    size_t pending = 0;
    std::vector<size_t> mine;                      // shards no worker could take: they run here, in order, BEFORE shard 0 returns
    for (size_t s = 1; s < shards; s++) {
        { std::lock_guard<std::mutex> lk(dm); pending++; }
        const bool ok = Workers::get().post([&, s] {
            try { fn(s); } catch (...) {}
            stats[s] = t_last;
            std::lock_guard<std::mutex> lk(dm);
            pending--;
            dcv.notify_all();
        });
        if (!ok) { { std::lock_guard<std::mutex> lk(dm); pending--; } mine.push_back(s); }
    }
    fn(0);
    stats[0] = t_last;
    for (size_t s : mine) { fn(s); stats[s] = t_last; }
    {
        std::unique_lock<std::mutex> lk(dm);
        dcv.wait(lk, [&] { return pending == 0; });
    }
    // the call's timing is that of its slowest shard
    for (size_t s = 1; s < shards; s++)
        if (stats[s].valid && stats[s].pipeline_ms > t_last.pipeline_ms) t_last = stats[s];
    std::lock_guard<std::mutex> lk(g_mu);
    g_last = t_last;
}

// Shards of a pipelined single-device call (msm_host_abi), from profiles/r03_h2d_pipeline.txt: G1 from 2^19 records in shards of
// about 350 000 (2^20: 2 / 3 / 4 shards 5.64 / 5.49 / 5.58 ms against 6.74 for one pipeline; below 2^19 a shard's pipeline is
// mostly fixed latency and nothing is gained), G2 -- three times the arithmetic for 1.8 times the bytes per record -- from 2^18
// in shards of 2^17 (2^18: 6.00 -> 5.52 ms; 2^17: no gain); at most 6 of the device's engine slots.
// EIP2537_H2D_PIPELINE=0: one copy, one pipeline; k >= 2: k shards whatever the size (A/B).
template <class F> static size_t pipeline_shards(size_t n) {
    static const int mode = [] { const char *v = getenv("EIP2537_H2D_PIPELINE"); return v ? atoi(v) : 1; }();
    size_t cap;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        cap = std::min<size_t>(6, (size_t)g_nslots);
    }
    if (mode == 0 || g_window_override.load() != 0) return 1;
    if (mode >= 2) return n >= (size_t)1024 * (size_t)mode ? std::min(cap, (size_t)mode) : 1;
    const bool g1 = sizeof(F) == sizeof(Fp);
    if (n < (g1 ? (size_t)1 << 19 : (size_t)1 << 18)) return 1;
    return std::min(cap, std::max<size_t>(2, g1 ? n / 349525 : n >> 17));
}
// Record shards of a staged G1 call (round 4; msm.hip runs them: the shards share one bucket space, so only the LAST shard's
// decode / sort / accumulate and one bucket reduce follow the last byte of the copy).  A small first shard starts the device after
// ~0.2 ms of copy instead of ~1 ms; the rest are equal shards of about 2^18 records -- below that a shard's sort and task kernels
// (which visit all 557 056 buckets whatever the shard holds) stop paying for the overlap (profiles/r04_h2d_stages.txt).
// $EIP2537_H2D_STAGES: "k" = k equal shards, or a comma list of relative shard weights (A/B); EIP2537_H2D_PIPELINE=0: one copy.
static ShardFeed stage_plan_g1(size_t n) {
    static const int mode = [] { const char *v = getenv("EIP2537_H2D_PIPELINE"); return v ? atoi(v) : 1; }();
    static const std::vector<uint32_t> env = parse_weights(getenv("EIP2537_H2D_STAGES"));
    if (mode == 0 || g_window_override.load() != 0 || n >= ((size_t)1 << 31)) return ShardFeed{};
    // (only the c = 16 plans, n > 2^17, share buckets: msm.hip takes any other call in one copy whatever is asked for here)
    std::vector<uint64_t> w;
    if (env.size() == 1) w.assign(std::min<size_t>(env[0], ShardFeed::kMax), 1u);
    else if (env.size() > 1) w.assign(env.begin(), env.begin() + std::min<size_t>(env.size(), ShardFeed::kMax));
    else {
        if (n < ((size_t)1 << 19)) return ShardFeed{};
        const size_t first = (size_t)1 << 16;
        const size_t rest = std::min<size_t>(ShardFeed::kMax - 1, std::max<size_t>(1, (n - first + ((size_t)1 << 17)) >> 18));
        w.push_back(first);
        for (size_t i = 0; i < rest; i++) w.push_back((n - first) / rest);
    }
    return feed_from_weights(n, w);
}
static int least_busy_pool() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!device_select_locked()) return -1;
    int pi = g_split[0];
    for (int i = 1; i < g_nsplit; i++)
        if (g_pools[g_split[i]].nbusy < g_pools[pi].nbusy) pi = g_split[i];
    return pi;
}
template <class F> static int msm_host_abi(byte *out, const byte *in, size_t in_len) {
    const size_t rec = Wire<F>::kMsmRecWords * 4;
    if (in_len == 0 || in_len % rec) return E_INVALID_LENGTH;      // before touching `in`
    const size_t n = in_len / rec;
    if (host_route(n, HostMax<F>::kUnits)) return device_present() ? msm_host_small<F>(out, in, n) : E_MEMORY_ERROR;
    if (n >= CoalesceCfg<F>::kMinRecords && n <= CoalesceCfg<F>::kMaxRecords && coalesce_enabled() && g_window_override.load() == 0)
        return msm_coalesced<F>(out, in, n);
    // Record ranges: one per listed device when the input is at least two shards' worth (split_plan); inside a device's range the
    // copy is hidden behind compute -- the 168 MB copy of 2^20 G1 records (3.2 ms at the link's ~53 GB/s) is as long as the whole
    // device pipeline and a sum over records can be cut anywhere.
    std::vector<int> pools = split_plan(n, SplitMin<F>::kRecords);
    if (std::is_same<F, Fp>::value) {
        // G1: a device's range is STAGED (stage_plan_g1: record shards copied back to back by the slot's helper thread, decoded /
        // sorted / accumulated behind their own copy into one shared bucket space; one bucket reduce, one host tail: msm.hip)
        if (pools.empty()) {
            const ShardFeed feed = stage_plan_g1(n);
            return msm_entry<F>(-1, out, in, n, false, false, nullptr, 0, &feed);
        }
        const size_t shards = pools.size();
        std::vector<Xyzz<F>> parts(shards);
        std::vector<int> rc(shards, E_MEMORY_ERROR);
        run_shards(shards, [&](size_t s) {
            const size_t lo = n * s / shards, hi = n * (s + 1) / shards;
            const ShardFeed feed = stage_plan_g1(hi - lo);
            rc[s] = msm_entry<F>(pools[s], reinterpret_cast<byte *>(&parts[s]), in + lo * rec, hi - lo, false, true, nullptr, 0, &feed);
        });
        for (size_t s = 0; s < shards; s++)
            if (rc[s]) return rc[s];
        Xyzz<F> acc = parts[0];
        for (size_t s = 1; s < shards; s++) acc = add(acc, parts[s]);
        host_encode_point<F>(out, to_affine(acc));
        return E_SUCCESS;
    }
    // G2 (c <= 13 plans up to 2^18 records: no shared bucket space): the round-3 form, independent shard pipelines
    if (pools.empty()) {
        if (pipeline_shards<F>(n) < 2) return msm_entry<F>(-1, out, in, n, false, false);
        const int pi = least_busy_pool();
        if (pi < 0) return E_MEMORY_ERROR;
        pools.assign(1, pi);
    }
    struct Shard { int pool; size_t lo, hi; CopyGate *gate; int turn; };
    std::vector<Shard> plan;
    std::vector<std::unique_ptr<CopyGate>> gates;
    for (size_t d = 0; d < pools.size(); d++) {
        const size_t lo = n * d / pools.size(), hi = n * (d + 1) / pools.size(), k = pipeline_shards<F>(hi - lo);
        gates.emplace_back(k > 1 ? new CopyGate : nullptr);
        for (size_t t = 0; t < k; t++)
            plan.push_back(Shard{pools[d], lo + (hi - lo) * t / k, lo + (hi - lo) * (t + 1) / k, gates.back().get(), (int)t});
    }
    const size_t shards = plan.size();
    std::vector<Xyzz<F>> parts(shards);
    std::vector<int> rc(shards, E_MEMORY_ERROR);
    run_shards(shards, [&](size_t s) {
        const Shard &sh = plan[s];
        rc[s] = msm_entry<F>(sh.pool, reinterpret_cast<byte *>(&parts[s]), in + sh.lo * rec, sh.hi - sh.lo, false, true, sh.gate, sh.turn);
        if (sh.gate) sh.gate->done(sh.turn);       // whatever happened: the device's next shard may copy
    });
    for (size_t s = 0; s < shards; s++)
        if (rc[s]) return rc[s];
    Xyzz<F> acc = parts[0];
    for (size_t s = 1; s < shards; s++) acc = add(acc, parts[s]);
    host_encode_point<F>(out, to_affine(acc));
    return E_SUCCESS;
}
template <class F> static int msm_combine(byte *out, const uint8_t *partials, size_t count) {
    Xyzz<F> acc = xyzz_inf<F>();
    for (size_t i = 0; i < count; i++) {
        Xyzz<F> p;
        memcpy(&p, partials + i * sizeof p, sizeof p);
        acc = add(acc, p);
    }
    host_encode_point<F>(out, to_affine(acc));
    return E_SUCCESS;
}

static void pairing_finish(byte *out, const Fp12 &ml) {
    bool one = is_one(final_exp_host(ml));
    memset(out, 0, 32);
    if (one) out[31] = 1;
}
static int pairing_entry(int pi, byte *out, const void *in, size_t k, bool device_input, bool want_partial) {
    SlotLease lease(pi);
    Engine *e = lease.e;
    if (!e) return E_MEMORY_ERROR;
    const void *d_in = in;
    if (!device_input) {
        int st = stage_input(e, in, k * 384);
        if (st) return st;
        d_in = e->input.p;
    }
    Fp12 ml;
    int st = pairing_device(e, d_in, k, reinterpret_cast<uint32_t *>(&ml));
    if (st) return st;
    if (want_partial) memcpy(out, &ml, sizeof ml); else pairing_finish(out, ml);
    return E_SUCCESS;
}
static int pairing_dev_abi(byte *out, const void *d_in, size_t k, bool want_partial) {
    if (!k) return E_INVALID_LENGTH;
    int pi = pool_of_pointer(d_in);
    if (pi == -2) { fprintf(stderr, "[eip2537_hip] %p is not device memory\n", d_in); return E_MEMORY_ERROR; }
    if (pi < 0) return E_MEMORY_ERROR;
    return pairing_entry(pi, out, d_in, k, true, want_partial);
}
// Concurrent small pairing checks are coalesced the same way: one decode / membership / line walk over the
// concatenated pairs (the walk costs the same ~1.2 ms for 8 pairs as for 2 000), one product-tree block per
// (call, step), and every caller finishes its own 63-squaring Horner pass and final exponentiation.
struct PairReq {
    const byte *in = nullptr;
    size_t n = 0;                          // pairs
    Fp12 L[kPairSteps];                    // per-step line products (batch) ...
    Fp12 ml;                               // ... or the whole Miller product (batch of one: the ordinary pipeline)
    bool have_ml = false;
    int rc = E_MEMORY_ERROR;
    bool taken = false, done = false;
};
static void run_pair_batch(std::vector<PairReq *> &batch) {
    const int M = (int)batch.size();
    SlotLease lease(-1);
    Engine *e = lease.e;
    if (!e) { for (auto *r : batch) r->rc = E_MEMORY_ERROR; return; }
    if (M == 1) {
        PairReq *r = batch[0];
        int st = stage_input(e, r->in, r->n * 384);
        if (!st) st = pairing_device(e, e->input.p, r->n, reinterpret_cast<uint32_t *>(&r->ml));
        r->have_ml = st == E_SUCCESS;
        r->rc = st;
        return;
    }
    std::vector<uint32_t> coff((size_t)M + 1, 0u);
    for (int j = 0; j < M; j++) coff[(size_t)j + 1] = coff[(size_t)j] + (uint32_t)batch[(size_t)j]->n;
    const size_t total = coff[(size_t)M];
    int st = e->input.reserve(total * 384) == hipSuccess ? E_SUCCESS : E_MEMORY_ERROR;
    for (int j = 0; j < M && !st; j++)
        if (hipMemcpyAsync(static_cast<char *>(e->input.p) + (size_t)coff[(size_t)j] * 384, batch[(size_t)j]->in,
                           batch[(size_t)j]->n * 384, hipMemcpyHostToDevice, e->stream) != hipSuccess) st = E_MEMORY_ERROR;
    std::vector<Fp12> L((size_t)M * kPairSteps);
    std::vector<int> rc((size_t)M, E_MEMORY_ERROR);
    if (st) e->failed = true;
    else st = pairing_batch_device(e, e->input.p, coff.data(), M, reinterpret_cast<uint32_t *>(L.data()), rc.data());
    for (int j = 0; j < M; j++) {
        PairReq *r = batch[(size_t)j];
        r->rc = st ? st : rc[(size_t)j];
        if (!r->rc) memcpy(r->L, &L[(size_t)j * kPairSteps], sizeof r->L);
    }
}
static int pairing_coalesced(byte *out, const byte *in, size_t k) {
    PairReq req;
    req.in = in;
    req.n = k;
    coalesce(req, (size_t)kPairBatchMaxCalls, (size_t)2048, [](std::vector<PairReq *> &batch) { run_pair_batch(batch); });
    if (req.rc) return req.rc;
    if (req.have_ml) { pairing_finish(out, req.ml); return E_SUCCESS; }
#if defined(EIP_HAVE_IFMA)
    if (host_ifma_enabled()) {                                   // the 68 per-step products by AVX-512 IFMA (ifma.h)
        int nsq[kPairSteps];
        uint64_t zbits = K_Z_ABS;
        int si = 0;
        for (int bit = 62; bit >= 0; bit--) { nsq[si++] = 1; if ((zbits >> bit) & 1ull) nsq[si++] = 0; }
        pairing_finish(out, conj(ifma::horner_groups_ifma(req.L, nsq, kPairSteps)));
        return E_SUCCESS;
    }
#endif
    pairing_finish(out, miller_product_from_steps(req.L));
    return E_SUCCESS;
}
static int pairing_host_abi(byte *out, const byte *in, size_t in_len) {
    if (in_len == 0 || in_len % 384) return E_INVALID_LENGTH;       // before touching `in`
    const size_t k = in_len / 384;
    if (host_route(k, kHostMaxPairs)) return device_present() ? pairing_host_small(out, in, k) : E_MEMORY_ERROR;
    if (k <= (size_t)kPairBatchMaxPairs && coalesce_enabled()) return pairing_coalesced(out, in, k);
    const std::vector<int> pools = split_plan(k, kPairingSplitMin);
    if (pools.empty()) return pairing_entry(-1, out, in, k, false, false);
    const size_t shards = pools.size();
    std::vector<Fp12> parts(shards);
    std::vector<int> rc(shards, E_MEMORY_ERROR);
    run_shards(shards, [&](size_t s) {
        const size_t lo = k * s / shards, hi = k * (s + 1) / shards;
        rc[s] = pairing_entry(pools[s], reinterpret_cast<byte *>(&parts[s]), in + lo * 384, hi - lo, false, true);
    });
    for (size_t s = 0; s < shards; s++)
        if (rc[s]) return rc[s];
    Fp12 acc = parts[0];
    for (size_t s = 1; s < shards; s++) acc = mul(acc, parts[s]);
    pairing_finish(out, acc);
    return E_SUCCESS;
}

static uint64_t msm_gas(uint64_t len, uint64_t rec, uint64_t mul_gas);

// ------------------------------------------------------------------ synthetic inputs (bench / tests)
// Records i in [start, start+n) of the SURVEY.md 8d workload: P_i = [a + i*b]G (G the group
// generator), k_i = four big-endian SplitMix64 outputs at stream position 4*i (uniform 256-bit,
// not reduced mod r -- the distribution of the reference bench, rust/benches/eip2537_benches.rs:61-63).
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t index) {
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class F> static Aff<F> group_generator();
template <> Aff<Fp> group_generator<Fp>() { return Aff<Fp>{Fp{{K_G1_X}}, Fp{{K_G1_Y}}}; }
template <> Aff<Fp2> group_generator<Fp2>() {
    return Aff<Fp2>{Fp2{Fp{{K_G2_X_C0}}, Fp{{K_G2_X_C1}}}, Fp2{Fp{{K_G2_Y_C0}}, Fp{{K_G2_Y_C1}}}};
}
template <class F>
static int gen_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32],
                         uint64_t seed, uint64_t start, size_t rec_bytes, bool with_scalars) {
    const size_t pb = Wire<F>::kPointWords * 4;
    uint32_t a[8], b[8];
    memcpy(a, a_le, 32);
    memcpy(b, b_le, 32);
    Aff<F> g = group_generator<F>();
    Xyzz<F> step = scalar_mul(g, b, 256);
    Xyzz<F> cur;
    {   // cur = [a]G + [start]([b]G), start as a 64-bit scalar
        Xyzz<F> acc = xyzz_inf<F>();
        for (int i = 63; i >= 0; i--) {
            acc = dbl(acc);
            if ((start >> i) & 1ull) acc = add(acc, step);
        }
        cur = add(scalar_mul(g, a, 256), acc);
    }
    std::vector<Xyzz<F>> pts(n);
    std::vector<F> pref(n);
    F run = f_one<F>();
    for (size_t i = 0; i < n; i++) {
        pts[i] = cur;
        pref[i] = run;
        if (!is_inf(cur)) run = mul(run, mul(cur.zz, cur.zzz));
        cur = add(cur, step);
    }
    F invr = inv(run);
    for (size_t i = n; i-- > 0;) {
        Aff<F> af{f_zero<F>(), f_zero<F>()};
        if (!is_inf(pts[i])) {
            F t = mul(invr, pref[i]);                    // 1 / (zz * zzz)
            invr = mul(invr, mul(pts[i].zz, pts[i].zzz));
            af.x = mul(pts[i].x, mul(t, pts[i].zzz));
            af.y = mul(pts[i].y, mul(t, pts[i].zz));
        }
        host_encode_point<F>(out + i * rec_bytes, af);
    }
    if (with_scalars) {
        for (size_t i = 0; i < n; i++) {
            uint8_t *k = out + i * rec_bytes + pb;
            for (int w = 0; w < 4; w++) {
                uint64_t v = splitmix64_at(seed, 4 * (start + i) + w);
                for (int bb = 0; bb < 8; bb++) k[8 * w + bb] = (uint8_t)(v >> (56 - 8 * bb));
            }
        }
    }
    return 0;
}

}  // namespace eip

using namespace eip;

// ------------------------------------------------------------------ reference ABI
API EIP2537_ERROR bls12_g1add(byte out[128], const byte in[256], size_t in_len) { return (EIP2537_ERROR)host_add<Fp>(out, in, in_len); }
API EIP2537_ERROR bls12_g1mul(byte out[128], const byte in[160], size_t in_len) { return (EIP2537_ERROR)host_mul<Fp>(out, in, in_len); }
API EIP2537_ERROR bls12_g2add(byte out[256], const byte in[512], size_t in_len) { return (EIP2537_ERROR)host_add<Fp2>(out, in, in_len); }
API EIP2537_ERROR bls12_g2mul(byte out[256], const byte in[288], size_t in_len) { return (EIP2537_ERROR)host_mul<Fp2>(out, in, in_len); }

API EIP2537_ERROR bls12_g1multiexp(byte out[128], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp>(out, in, in_len); }
API EIP2537_ERROR bls12_g1multiexp_naive(byte out[128], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp>(out, in, in_len); }
API EIP2537_ERROR bls12_g1multiexp_bc(byte out[128], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp>(out, in, in_len); }
API EIP2537_ERROR bls12_g2multiexp(byte out[256], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp2>(out, in, in_len); }
API EIP2537_ERROR bls12_g2multiexp_naive(byte out[256], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp2>(out, in, in_len); }
API EIP2537_ERROR bls12_g2multiexp_bc(byte out[256], byte *in, size_t in_len) { return (EIP2537_ERROR)msm_host_abi<Fp2>(out, in, in_len); }

API EIP2537_ERROR bls12_pairing(byte out[32], byte *in, size_t in_len) { return (EIP2537_ERROR)pairing_host_abi(out, in, in_len); }

// map-to-curve: RFC 9380 map_to_curve of ONE field element, then cofactor clearing -- what the
// reference gets from blst_map_to_g1/_g2(out, u, NULL) (:1113, :1155).  Host code (csrc/h2c.h).
API EIP2537_ERROR bls12_map_fp_to_g1(byte out[128], const byte in[64], size_t in_len) {
    if (in_len != 64) return EIP2537_INVALID_LENGTH;
    uint32_t w[16];
    memcpy(w, in, 64);
    Fp u;
    if (fp_decode(u, w) < 0) return EIP2537_INVALID_ELEMENT;
    Aff<Fp> q = map_to_curve<Fp>(u);
    host_encode_point<Fp>(out, to_affine(scalar_mul(q, K_ISO_H_EFF_G1, 64)));          // h_eff = 1 - z
    return EIP2537_SUCCESS;
}
API EIP2537_ERROR bls12_map_fp2_to_g2(byte out[256], const byte in[128], size_t in_len) {
    if (in_len != 128) return EIP2537_INVALID_LENGTH;
    uint32_t w[32];
    memcpy(w, in, 128);
    Fp2 u;
    if (fp_decode(u, w) < 0) return EIP2537_INVALID_ELEMENT;
    Aff<Fp2> q = map_to_curve<Fp2>(u);
    static_assert(K_ISO_H_EFF_G2_BITS <= 32 * 20, "cofactor words");
    host_encode_point<Fp2>(out, to_affine(msm_interleaved<Fp2>(&q, K_ISO_H_EFF_G2, 1, 20)));    // h2 (3 z^2 - 3): 636 bits, signed 5-bit windows
    return EIP2537_SUCCESS;
}

// ------------------------------------------------------------------ gas (reference :1168-1271)
extern "C" {
__attribute__((visibility("default"))) extern const uint64_t BLS12_G1ADD_GAS = 600;
__attribute__((visibility("default"))) extern const uint64_t BLS12_G1MUL_GAS = 12000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_G2ADD_GAS = 4500;
__attribute__((visibility("default"))) extern const uint64_t BLS12_G2MUL_GAS = 55000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_PAIRING_BASE_GAS = 115000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_PAIRING_PAIR_GAS = 23000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_MAP_FP_TO_G1_GAS = 5500;
__attribute__((visibility("default"))) extern const uint64_t BLS12_MAP_FP2_TO_G2_GAS = 110000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_MULTIEXP_MULTIPLIER_GAS = 1000;
__attribute__((visibility("default"))) extern const uint64_t BLS12_MULTIEXP_DISCOUNT_TABLE_LEN = 128;
__attribute__((visibility("default"))) extern const uint64_t BLS12_MULTIEXP_DISCOUNT[128] = {
    1200, 888, 764, 641, 594, 547, 500, 453, 438, 423, 408, 394, 379, 364, 349, 334,
    330, 326, 322, 318, 314, 310, 306, 302, 298, 294, 289, 285, 281, 277, 273, 269,
    268, 266, 265, 263, 262, 260, 259, 257, 256, 254, 253, 251, 250, 248, 247, 245,
    244, 242, 241, 239, 238, 236, 235, 233, 232, 231, 229, 228, 226, 225, 223, 222,
    221, 220, 219, 219, 218, 217, 216, 216, 215, 214, 213, 213, 212, 211, 211, 210,
    209, 208, 208, 207, 206, 205, 205, 204, 203, 202, 202, 201, 200, 199, 199, 198,
    197, 196, 196, 195, 194, 193, 193, 192, 191, 191, 190, 189, 188, 188, 187, 186,
    185, 185, 184, 183, 182, 182, 181, 180, 179, 179, 178, 177, 176, 176, 175, 174};
}
static uint64_t eip::msm_gas(uint64_t len, uint64_t rec, uint64_t mul_gas) {
    uint64_t k = len / rec;
    if (k == 0) return 0;
    uint64_t d = BLS12_MULTIEXP_DISCOUNT[k < BLS12_MULTIEXP_DISCOUNT_TABLE_LEN ? k - 1 : BLS12_MULTIEXP_DISCOUNT_TABLE_LEN - 1];
    return k * mul_gas * d / BLS12_MULTIEXP_MULTIPLIER_GAS;
}
API uint64_t bls12_g1add_gas(void) { return BLS12_G1ADD_GAS; }
API uint64_t bls12_g1mul_gas(void) { return BLS12_G1MUL_GAS; }
API uint64_t bls12_g1multiexp_gas(uint64_t len) { return msm_gas(len, 160, BLS12_G1MUL_GAS); }
API uint64_t bls12_g2add_gas(void) { return BLS12_G2ADD_GAS; }
API uint64_t bls12_g2mul_gas(void) { return BLS12_G2MUL_GAS; }
API uint64_t bls12_g2multiexp_gas(uint64_t len) { return msm_gas(len, 288, BLS12_G2MUL_GAS); }
API uint64_t bls12_pairing_gas(uint64_t len) {
    uint64_t k = len / 384;
    return k ? k * BLS12_PAIRING_PAIR_GAS + BLS12_PAIRING_BASE_GAS : 0;
}
API uint64_t bls12_map_fp_to_g1_gas(void) { return BLS12_MAP_FP_TO_G1_GAS; }
API uint64_t bls12_map_fp2_to_g2_gas(void) { return BLS12_MAP_FP2_TO_G2_GAS; }

// ------------------------------------------------------------------ extensions
API int eip2537_hip_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_nsplit) {           // already selected: fine if it is the same single device (or "whatever is selected")
        if (device < 0 || (g_nsplit == 1 && g_pools[g_split[0]].ordinal == device)) return 0;
        fprintf(stderr, "[eip2537_hip] eip2537_hip_init(%d): devices were already selected\n", device);
        return E_MEMORY_ERROR;
    }
    g_device_request = device;
    bool ok = device_select_locked();
    if (!ok) g_device_request = -1;
    return ok ? 0 : E_MEMORY_ERROR;
}
// Start-up hook for embedders (and the static shim, which exports the same name): ONCE from main(), before the process starts
// threads and before its first HIP call -- the only place where the environment can be touched safely.  Never overrides the
// embedder's own value.  eip2537_hip_hw_queues() reports what the runtime will see / has seen.
API int eip2537_hip_early_init(void) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    return 0;
}
API int eip2537_hip_hw_queues(void) {
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    return v && atoi(v) > 0 ? atoi(v) : 4;
}
API int eip2537_hip_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return device_select_locked() ? g_nsplit : 0;
}
API int eip2537_hip_g1multiexp_dev(uint8_t out[128], const void *d_in, size_t n) { return msm_dev_abi<Fp>(out, d_in, n, false); }
API int eip2537_hip_g2multiexp_dev(uint8_t out[256], const void *d_in, size_t n) { return msm_dev_abi<Fp2>(out, d_in, n, false); }
API int eip2537_hip_pairing_dev(uint8_t out[32], const void *d_in, size_t k) { return pairing_dev_abi(out, d_in, k, false); }
API int eip2537_hip_g1msm_partial_dev(uint8_t *partial, const void *d_in, size_t n) { return msm_dev_abi<Fp>(partial, d_in, n, true); }
API int eip2537_hip_g2msm_partial_dev(uint8_t *partial, const void *d_in, size_t n) { return msm_dev_abi<Fp2>(partial, d_in, n, true); }
API int eip2537_hip_pairing_partial_dev(uint8_t *partial, const void *d_in, size_t k) { return pairing_dev_abi(partial, d_in, k, true); }
API int eip2537_hip_g1msm_combine(uint8_t out[128], const uint8_t *partials, size_t count) { return msm_combine<Fp>(out, partials, count); }
API int eip2537_hip_g2msm_combine(uint8_t out[256], const uint8_t *partials, size_t count) { return msm_combine<Fp2>(out, partials, count); }
API int eip2537_hip_pairing_combine(uint8_t out[32], const uint8_t *partials, size_t count) {
    Fp12 acc = fp12_one();
    for (size_t i = 0; i < count; i++) {
        Fp12 p;
        memcpy(&p, partials + i * sizeof p, sizeof p);
        acc = mul(acc, p);
    }
    pairing_finish(out, acc);
    return 0;
}
API void eip2537_hip_last_timing(float *pipeline_ms, float *dominant_kernel_ms) {
    // the calling thread's last call if it made one, else the process's last call
    std::lock_guard<std::mutex> lk(g_mu);
    const CallStats &c = t_last.valid ? t_last : g_last;
    if (pipeline_ms) *pipeline_ms = c.pipeline_ms;
    if (dominant_kernel_ms) *dominant_kernel_ms = c.dominant_ms;
}
API void eip2537_hip_last_timing_aux(float *aux1_ms, float *aux2_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    const CallStats &c = t_last.valid ? t_last : g_last;
    if (aux1_ms) *aux1_ms = c.aux_ms[0];
    if (aux2_ms) *aux2_ms = c.aux_ms[1];
}
API int eip2537_hip_last_plan(char *kernel_name, size_t cap, int *window_bits, int *windows, int *lanes,
                              uint32_t *units, uint32_t *buckets) {
    std::lock_guard<std::mutex> lk(g_mu);
    const CallStats &c = t_last.valid ? t_last : g_last;
    if (!c.valid) return E_EMPTY_INPUT;
    if (kernel_name && cap) { strncpy(kernel_name, c.plan.kernel, cap - 1); kernel_name[cap - 1] = 0; }
    if (window_bits) *window_bits = c.plan.c;
    if (windows) *windows = c.plan.windows;
    if (lanes) *lanes = c.plan.lanes;
    if (units) *units = c.plan.units;
    if (buckets) *buckets = c.plan.buckets;
    return 0;
}
API int eip2537_hip_last_shards(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    const CallStats &c = t_last.valid ? t_last : g_last;
    return c.valid ? c.plan.shards : 0;
}
// Release the workspace of every idle engine slot that holds more than keep_bytes (slots grow to the
// largest call they have served: ~0.9 GB after one 2^20-record MSM).  Returns the bytes released.
API size_t eip2537_hip_trim(size_t keep_bytes) {
    size_t freed = 0;
    std::unique_lock<std::mutex> lk(g_mu);
    for (int p = 0; p < g_npools; p++)
        for (int i = 0; i < kMaxSlots; i++) {
            Engine &e = g_pools[p].slots[i];
            if (g_pools[p].busy[i] || !e.ready) continue;
            const size_t ws = e.workspace_bytes();
            if (ws <= keep_bytes) continue;
            g_pools[p].busy[i] = true;           // ours while the lock is dropped for the frees
            lk.unlock();
            { DeviceGuard g(g_pools[p].ordinal); e.release_workspace(); }
            lk.lock();
            g_pools[p].busy[i] = false;
            freed += ws;
        }
    g_cv.notify_all();
    return freed;
}
API void eip2537_hip_coalesce_stats(uint64_t *pipelines, uint64_t *calls, uint64_t *largest_batch) {
    if (pipelines) *pipelines = g_co_batches.load();
    if (calls) *calls = g_co_calls.load();
    if (largest_batch) *largest_batch = g_co_max.load();
}
API int eip2537_hip_set_route(int route) {
    if (route < -1 || route > 1) return E_INVALID_LENGTH;
    g_route_override.store(route);
    return 0;
}
API int eip2537_hip_set_window(int c) {
    if (c != 0 && (c < 4 || c > 16)) return E_INVALID_LENGTH;
    g_window_override.store(c);
    return 0;
}

// EIP_HOST_ONLY: the ThreadSanitizer build of this file's host concurrency code (tools/tsan/) has no device code at all
#ifndef EIP_HOST_ONLY
// Device self-test of the field products: n pseudo-random operand pairs (every eighth pair from a table
// of extreme values) through the column products the kernels use -- canonical and lazy, product and
// square -- against the independent 12 x 32-bit CIOS product, all on the device. Guards the products
// against code generation differences between host and device (field.h, radix 2^30 note).
__device__ static Fp selftest_operand(uint64_t &s, unsigned idx) {
    Fp v;
    for (int k = 0; k < 12; k += 2) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v.l[k] = (uint32_t)s;
        v.l[k + 1] = (uint32_t)(s >> 32);
    }
    if ((idx & 7) == 7) {
        const unsigned kind = (idx >> 3) % 6;
        const uint32_t p[12] = {K_P};
        for (int k = 0; k < 12; k++)
            v.l[k] = kind == 0 ? 0u : kind == 1 ? 0xffffffffu : kind == 2 ? p[k] : kind == 3 ? (k == 0 ? 1u : 0u)
                   : kind == 4 ? (k & 1 ? 0xffffffffu : 0u) : (v.l[k] | 0x3fffffffu);
        if (kind == 2) v.l[0] -= 1 + ((idx >> 6) & 1);          // p - 1, p - 2
    }
    v.l[11] &= 0x0fffffffu;                                     // below 2^380 < 2p: legal for every product
    return v;
}
__global__ void k_field_selftest(uint64_t seed, unsigned n, unsigned long long *bad) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = (seed ^ (0x9E3779B97F4A7C15ull * (i + 1))) | 1;
    const Fp a = selftest_operand(s, i), b = selftest_operand(s, i + 3);
    const Fp want = fp_mul_limbs32(a, b), want2 = fp_mul_limbs32(a, a);
    if (!eq(fp_mul_cols_t<true>(a, b), want)) atomicAdd(&bad[0], 1ull);
    if (!eq(fp_sqr_cols_t<true>(a), want2)) atomicAdd(&bad[1], 1ull);
    if (!eq(fp_canon(mul(FpI{a}, FpI{b})), want)) atomicAdd(&bad[2], 1ull);
    if (!eq(fp_canon(sqr(FpI{a})), want2)) atomicAdd(&bad[3], 1ull);
}
API int eip2537_hip_field_selftest(uint64_t seed, size_t n, uint64_t mismatches[4]) {
    if (!mismatches || n == 0 || n > (1u << 24)) return E_INVALID_LENGTH;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!device_select_locked()) return E_MEMORY_ERROR;
    }
    DeviceGuard guard(g_pools[g_split[0]].ordinal);
    if (!guard.ok) return E_MEMORY_ERROR;
    unsigned long long *d_bad = nullptr;
    if (hipMalloc(&d_bad, 4 * sizeof(unsigned long long)) != hipSuccess) return E_MEMORY_ERROR;
    unsigned long long h_bad[4] = {0, 0, 0, 0};
    bool ok = hipMemset(d_bad, 0, sizeof(h_bad)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_field_selftest, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, seed, (unsigned)n, d_bad);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_bad);
    if (!ok) return E_MEMORY_ERROR;
    for (int k = 0; k < 4; k++) mismatches[k] = h_bad[k];
    return 0;
}

// Device self-test of the LIMB-FORM primitives (limb30.h, limbk.h, fp_mul2_cols30): the G1 accumulate / fold /
// reduce and every pairing kernel compute on them, and tools/limb30_check.hip / pairing_limb_check.hip run the same
// source on the HOST only -- the 13 x 30-bit column product has a record of device-only code generation trouble
// (field.h).  Everything is checked against the independent 12 x 32-bit CIOS product, on the device, with operands
// grown to the bounds the kernels use (8 p, 10 p, 25 p, 600 p) and the 24-bit top limb that to_limbs() produces.
//   [0] mulL  [1] sqrL  [2] mul2L on grown operands  [3] fp_mul2_cols30  [4] to_limbs / to_fpi round trip
//   [5] weak_reduceL / shlL  [6] subL / sub2L / addL / dbl_addL / negL  [7] is_zero_modp
__device__ static FpL selftest_grow(const FpL &a, uint32_t k) {           // a + k p
    uint32_t p1[13];
    kp30<1>(p1);
    FpL pl, r = a;
    for (int i = 0; i < 13; i++) pl.l[i] = p1[i];
    for (uint32_t e = 0; e < k; e++) r = addL(r, pl);
    return r;
}
__global__ void k_limb_selftest(uint64_t seed, unsigned n, unsigned long long *bad) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = (seed ^ (0x9E3779B97F4A7C15ull * (i + 1))) | 1;
    // canonical Montgomery values (factor 2^384) from arbitrary operands
    const Fp one_m{{K_ONE}}, r390{{K_R390_MODP}};
    const Fp a = fp_mul_limbs32(selftest_operand(s, i), one_m), b = fp_mul_limbs32(selftest_operand(s, i + 3), one_m);
    const Fp c = fp_mul_limbs32(selftest_operand(s, i + 5), one_m), d = fp_mul_limbs32(selftest_operand(s, i + 6), one_m);
    auto lift = [&](const Fp &v) { return to_limbs(fp_mul_limbs32(v, r390)); };                    // v R -> v R' (top limb: 24 bits)
    auto canon_of = [&](const FpL &v) { return fp_reduce_once(to_fpi(v).v); };                     // v R' (< 600 p) -> v R canonical
    auto mul32 = [&](const Fp &x, const Fp &y) { return fp_mul_limbs32(x, y); };
    const FpL A = lift(a), B = lift(b), C = lift(c), D = lift(d);
    const uint32_t grow[4] = {0u, 7u, 9u, 24u};
    const FpL Ag = selftest_grow(A, grow[i & 3]), Bg = selftest_grow(B, grow[(i >> 2) & 3]);
    if (!eq(canon_of(mulL(Ag, Bg)), mul32(a, b))) atomicAdd(&bad[0], 1ull);
    if (!eq(canon_of(sqrL(Ag)), mul32(a, a))) atomicAdd(&bad[1], 1ull);
    {
        const FpL g1 = subL<8>(A, B), g2 = subL<6>(B, A);                        // bounds 9 x 7, plus a second product: < 630
        const Fp want = sub(mul32(sub(a, b), sub(b, a)), mul32(a, b));
        if (!eq(canon_of(mul2L(g1, g2, A, negL<2>(B))), want)) atomicAdd(&bad[2], 1ull);
        // six products, one reduction (round 4: the fused line product of the pairing fold)
        const Fp want6 = add(add(want, add(mul32(a, a), mul32(b, sub(a, b)))), add(mul32(sub(b, a), sub(b, a)), mul32(a, b)));
        if (!eq(canon_of(mul6L(g1, g2, A, negL<2>(B), A, A, B, g1, g2, g2, A, B)), want6)) atomicAdd(&bad[2], 1ull);
        // four products, one reduction (the G2 accumulate's Y3 per component)
        if (!eq(canon_of(mul4L(g1, g2, A, negL<2>(B), A, A, B, g1)), add(want, add(mul32(a, a), mul32(b, sub(a, b)))))) atomicAdd(&bad[2], 1ull);
        const FpL big = selftest_grow(C, 590u);                                  // the largest operands limbk.h admits
        if (!eq(canon_of(mulL(big, D)), mul32(c, d))) atomicAdd(&bad[2], 1ull);
    }
    if (!eq(fp_reduce_once(fp_mul2_cols30(a, b, c, d)), add(mul32(a, b), mul32(c, d)))) atomicAdd(&bad[3], 1ull);
    if (i < 64) {
        // all-ones limbs (x = 2^384 - 1 < 10 p): the operand that fills the 64-bit columns the most -- the carry-out schedules of
        // limb30.h / field.h are sized for it (tools/limb_column_bounds.py), and the device carries differ from the host's
        // (col_carry_hi is an explicit v_mad_u64_u32 here).  x^2 / R through the limb square, the general product and the
        // two-product sum against the 32-bit CIOS product of x mod p.
        FpL x;
#pragma unroll
        for (int k = 0; k < 12; k++) x.l[k] = kM30;
        x.l[12] = 0x00ffffffu;
        Fp v = from_limbs(x);
        for (int r = 0; r < 12; r++) v = fp_reduce_once(v);
        const Fp want = mul32(v, v), sq = canon_of(sqrL(x));
        if (!eq(mul32(mul32(sq, r390), r390), want)) atomicAdd(&bad[1], 1ull);
        if (!eq(canon_of(mulL(x, x)), sq)) atomicAdd(&bad[0], 1ull);
        if (!eq(canon_of(mul2L(x, x, x, x)), add(sq, sq))) atomicAdd(&bad[2], 1ull);
        if (!eq(canon_of(mul6L(x, x, x, x, x, x, x, x, x, x, x, x)), add(add(add(sq, sq), add(sq, sq)), add(sq, sq)))) atomicAdd(&bad[2], 1ull);
        if (!eq(canon_of(mul4L(x, x, x, x, x, x, x, x)), add(add(sq, sq), add(sq, sq)))) atomicAdd(&bad[2], 1ull);
        Fp w;
#pragma unroll
        for (int k = 0; k < 12; k++) w.l[k] = k == 11 ? 0x0fffffffu : 0xffffffffu;      // < 2^380: inside fp_mul2_cols30's [0, 2p) range
        const Fp wr = fp_reduce_once(w);
        if (!eq(fp_reduce_once(fp_mul2_cols30(w, w, w, w)), dbl(mul32(wr, wr)))) atomicAdd(&bad[3], 1ull);
    }
    if (!eq(canon_of(A), a) || !eq(from_limbs(to_limbs(a)), a)) atomicAdd(&bad[4], 1ull);
    {
        const FpL big = selftest_grow(A, 100u + (i % 490u));
        const FpL w = weak_reduceL(big);
        const FpL room = negL<3>(w);                                             // 3 p - w must not be negative
        if (!eq(canon_of(w), a) || (int32_t)room.l[12] < 0) atomicAdd(&bad[5], 1ull);
        if (!eq(canon_of(shlL<3>(Ag)), dbl(dbl(dbl(a)))) || !eq(canon_of(shlL<1>(Ag)), dbl(a))) atomicAdd(&bad[5], 1ull);
    }
    if (!eq(canon_of(subL<2>(A, B)), sub(a, b)) || !eq(canon_of(sub2L<4>(A, B)), sub(a, dbl(b))) || !eq(canon_of(addL(Ag, Bg)), add(a, b)) ||
        !eq(canon_of(dbl_addL(A, B)), add(dbl(a), b)) || !eq(canon_of(negL<2>(B)), neg(b)))
        atomicAdd(&bad[6], 1ull);
    if (is_zero_modp(subL<8>(A, B), 10) != eq(a, b) || !is_zero_modp(subL<8>(A, A), 10) || !is_zero_modp(selftest_grow(subL<3>(A, A), 20u), 25))
        atomicAdd(&bad[7], 1ull);
}
API int eip2537_hip_limb_selftest(uint64_t seed, size_t n, uint64_t mismatches[8]) {
    if (!mismatches || n == 0 || n > (1u << 24)) return E_INVALID_LENGTH;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!device_select_locked()) return E_MEMORY_ERROR;
    }
    DeviceGuard guard(g_pools[g_split[0]].ordinal);
    if (!guard.ok) return E_MEMORY_ERROR;
    unsigned long long *d_bad = nullptr;
    if (hipMalloc(&d_bad, 8 * sizeof(unsigned long long)) != hipSuccess) return E_MEMORY_ERROR;
    unsigned long long h_bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = hipMemset(d_bad, 0, sizeof(h_bad)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_limb_selftest, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, seed, (unsigned)n, d_bad);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d_bad);
    if (!ok) return E_MEMORY_ERROR;
    for (int k = 0; k < 8; k++) mismatches[k] = h_bad[k];
    return 0;
}

#else
API int eip2537_hip_field_selftest(uint64_t, size_t, uint64_t *) { return E_MEMORY_ERROR; }
API int eip2537_hip_limb_selftest(uint64_t, size_t, uint64_t *) { return E_MEMORY_ERROR; }
#endif

// Synthetic workloads of SURVEY.md 8d (host code; used by bench.py and the tests, never by a precompile)
API int eip2537_hip_gen_g1_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start) {
    return gen_msm_input<Fp>(out, n, a_le, b_le, seed, start, 160, true);
}
API int eip2537_hip_gen_g2_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start) {
    return gen_msm_input<Fp2>(out, n, a_le, b_le, seed, start, 288, true);
}
// pairs ([a0 + i*a1]G1, [b0 + i*b1]G2), i in [start, start+k)
API int eip2537_hip_gen_pairing_input(uint8_t *out, size_t k, const uint8_t a0[32], const uint8_t a1[32],
                                      const uint8_t b0[32], const uint8_t b1[32], uint64_t start) {
    gen_msm_input<Fp>(out, k, a0, a1, 0, start, 384, false);
    return gen_msm_input<Fp2>(out + 128, k, b0, b1, 0, start, 384, false);
}
