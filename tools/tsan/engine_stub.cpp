// Stand-ins for the device pipelines of msm.hip / pairing.hip, for the ThreadSanitizer build of the library's host
// concurrency code (tests/test_host_tsan.py).  They read the staged records from the stub "device" memory, apply the
// real wire validation (so that error paths are exercised) and return CHEAP deterministic values with the algebraic
// shape the callers rely on -- an MSM partial is a sum over records (so shards add up, and window sums recombine by
// Horner), a Miller "product" is a product over pairs -- instead of the real sums, which would take seconds on a CPU.
// Never linked into the product.
#include <vector>
#include "../../blst_eip2537_amd/csrc/codec.h"
#include "../../blst_eip2537_amd/csrc/pairing.h"
#include "../../blst_eip2537_amd/csrc/engine.h"

namespace eip {

MsmPlan msm_make_plan(uint32_t n, int, bool) { MsmPlan p{}; p.n = n; p.c = 8; p.W = 32; return p; }

// value of a record: [k mod 2^8] P  (one window of the scalar: enough for a sum that depends on every record)
template <class F> static int msm_stub(Engine *e, const void *d_in, size_t n, Xyzz<F> *acc_out, Xyzz<F> *wins) {
    const ShardFeed feed = e->feed;
    e->feed.k = 0;
    StagedCopy staged;                                        // as in msm.hip: declared before anything that can return
    if (e->host_src && feed.k > 1 && sizeof(F) == sizeof(Fp)) {
        // staged call (msm.hip): the slot's helper thread copies the shards, this thread waits for each shard's hand-over
        // before it "launches" (here: reads) that shard
        for (int sh = 0; sh < feed.k; sh++)
            if (!e->ev_copy[sh] && hipEventCreateWithFlags(&e->ev_copy[sh], hipEventDisableTiming) != hipSuccess) return E_MEMORY_ERROR;
        if (e->need_stream2() != hipSuccess) return E_MEMORY_ERROR;
        const void *src = e->host_src;
        e->host_src = nullptr;
        const bool threaded = staged.start(e->helper, e->device, e->stream2, e->ev_copy, feed, Wire<F>::kMsmRecWords * 4, e->input.p, src);
        for (int sh = 0; sh < feed.k; sh++) {
            if (threaded) { if (!staged.wait_shard(sh)) return E_MEMORY_ERROR; }
            else memcpy(static_cast<char *>(e->input.p) + (size_t)feed.bound[sh] * Wire<F>::kMsmRecWords * 4,
                        static_cast<const char *>(src) + (size_t)feed.bound[sh] * Wire<F>::kMsmRecWords * 4,
                        (size_t)(feed.bound[sh + 1] - feed.bound[sh]) * Wire<F>::kMsmRecWords * 4);
        }
        staged.finish();
    } else if (e->host_src) {                                 // host input: the real pipeline stages it itself (msm.hip)
        memcpy(e->input.p, e->host_src, n * Wire<F>::kMsmRecWords * 4);
        if (e->copy_gate) e->copy_gate->done(e->copy_turn);  // a shard of a pipelined call: the next shard may copy (msm.hip)
        e->host_src = nullptr;
    }
    const uint32_t *in = static_cast<const uint32_t *>(d_in);
    Xyzz<F> acc = xyzz_inf<F>();
    if (wins) for (int w = 0; w < kMsmBatchWindows; w++) wins[w] = xyzz_inf<F>();
    for (size_t i = 0; i < n; i++) {
        const uint32_t *w = in + i * Wire<F>::kMsmRecWords;
        Aff<F> a;
        const int st = decode_point<F>(a, w);
        if (st) return st;
        uint32_t k[8];
        decode_scalar(k, w + Wire<F>::kPointWords);
        const uint32_t small[1] = {k[0] & 0xffu};
        const Xyzz<F> t = scalar_mul(a, small, 8);
        acc = add(acc, t);
        if (wins) wins[0] = add(wins[0], t);                  // Horner over the 32 window sums gives the same value
    }
    (void)e;
    if (acc_out) *acc_out = acc;
    return E_SUCCESS;
}
int msm_g1_device(Engine *e, const void *d_in, size_t n, uint32_t *pw, int) { return msm_stub<Fp>(e, d_in, n, reinterpret_cast<Xyzz<Fp> *>(pw), nullptr); }
int msm_g2_device(Engine *e, const void *d_in, size_t n, uint32_t *pw, int) { return msm_stub<Fp2>(e, d_in, n, reinterpret_cast<Xyzz<Fp2> *>(pw), nullptr); }
template <class F> static int msm_batch_stub(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *wins_words, int *rc) {
    Xyzz<F> *wins = reinterpret_cast<Xyzz<F> *>(wins_words);
    for (int j = 0; j < M; j++)
        rc[j] = msm_stub<F>(e, static_cast<const uint32_t *>(d_in) + (size_t)coff[j] * Wire<F>::kMsmRecWords, coff[j + 1] - coff[j], nullptr,
                            wins + (size_t)j * kMsmBatchWindows);
    return E_SUCCESS;
}
int msm_g1_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *ww, int *rc) { return msm_batch_stub<Fp>(e, d_in, coff, M, ww, rc); }
int msm_g2_batch_device(Engine *e, const void *d_in, const uint32_t *coff, int M, uint32_t *ww, int *rc) { return msm_batch_stub<Fp2>(e, d_in, coff, M, ww, rc); }

// "Miller value" of a pair: an Fp12 built from the coordinates (a product over pairs, like the real thing)
static int pair_value(Fp12 &v, const uint32_t *w) {
    Aff<Fp> P;
    Aff<Fp2> Q;
    int st = decode_point<Fp>(P, w);
    if (st) return st;
    st = decode_point<Fp2>(Q, w + 32);
    if (st) return st;
    v = fp12_one();
    if (eq(P.x, Fp{{K_G1_X}}) && eq(P.y, Fp{{K_G1_Y}})) return E_SUCCESS;       // (G1, *) counts as one: a check of such pairs is "true"
    v.c0.a1 = Fp2{P.x, P.y};
    v.c1.a0 = Q.x;
    v.c1.a2 = Q.y;
    return E_SUCCESS;
}
int pairing_device(Engine *, const void *d_in, size_t k, uint32_t *ml_words) {
    Fp12 F = fp12_one();
    for (size_t i = 0; i < k; i++) {
        Fp12 v;
        const int st = pair_value(v, static_cast<const uint32_t *>(d_in) + i * 96);
        if (st) return st;
        F = mul(F, v);
    }
    memcpy(ml_words, &F, sizeof F);
    return E_SUCCESS;
}
int pairing_batch_device(Engine *, const void *d_in, const uint32_t *coff, int M, uint32_t *L_words, int *rc) {
    Fp12 *L = reinterpret_cast<Fp12 *>(L_words);
    for (int j = 0; j < M; j++) {
        Fp12 F = fp12_one();
        rc[j] = E_SUCCESS;
        for (uint32_t i = coff[j]; i < coff[j + 1] && !rc[j]; i++) {
            Fp12 v;
            rc[j] = pair_value(v, static_cast<const uint32_t *>(d_in) + (size_t)i * 96);
            if (!rc[j]) F = mul(F, v);
        }
        for (int s = 0; s < kPairSteps; s++) L[(size_t)j * kPairSteps + s] = s == kPairSteps - 1 ? F : fp12_one();   // Horner leaves F
    }
    return E_SUCCESS;
}

}  // namespace eip
