// Hammer for the host concurrency code of csrc/api.hip under ThreadSanitizer (tests/test_host_tsan.py): many threads call
// the reference ABI at once -- small multiexps and pairing checks (the coalescing queue: Batcher / coalesce), mid-size
// ones (engine slots: SlotLease, with only 2 slots so that callers wait), large ones (record-range split over two stub
// devices: run_shards) -- while other threads read the last-call statistics and trim idle workspaces.  Every result is
// compared with the one the same call returns on an otherwise idle library (taken before the threads start).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "../../include/eip2537.h"
#include "../../include/eip2537_hip.h"

struct Case { std::vector<uint8_t> in; std::vector<uint8_t> want; int rc; int kind; };   // kind 0 g1 msm, 1 g2 msm, 2 pairing

static int call(const Case &c, uint8_t *out) {
    if (c.kind == 0) return bls12_g1multiexp(out, const_cast<uint8_t *>(c.in.data()), c.in.size());
    if (c.kind == 1) return bls12_g2multiexp(out, const_cast<uint8_t *>(c.in.data()), c.in.size());
    return bls12_pairing(out, const_cast<uint8_t *>(c.in.data()), c.in.size());
}

int main() {
    uint8_t a[32] = {7}, b[32] = {3}, one[32] = {1}, zero[32] = {0};
    std::vector<Case> cases;
    auto add_msm = [&](int kind, size_t n, uint64_t seed, bool corrupt) {
        Case c;
        c.kind = kind;
        c.in.resize(n * (kind ? 288 : 160));
        if (kind) eip2537_hip_gen_g2_msm_input(c.in.data(), n, a, b, seed, 0);
        else eip2537_hip_gen_g1_msm_input(c.in.data(), n, a, b, seed, 0);
        if (corrupt) c.in[(n / 2) * (kind ? 288 : 160) + 70] ^= 0x55;       // a coordinate byte: not on the curve any more
        cases.push_back(c);
    };
    auto add_pairing = [&](size_t k, bool good) {
        Case c;
        c.kind = 2;
        c.in.resize(k * 384);
        eip2537_hip_gen_pairing_input(c.in.data(), k, one, zero, one, zero, 0);          // k x (G1, G2)
        if (!good) eip2537_hip_gen_pairing_input(c.in.data() + (k - 1) * 384, 1, a, zero, one, zero, 0);
        cases.push_back(c);
    };
    for (size_t n : {20, 33, 64, 100, 300, 512}) { add_msm(0, n, n, false); add_msm(1, n / 2 + 9, n, false); }   // coalesced
    add_msm(0, 40, 5, true);                                                                                    // an error inside a batch
    for (size_t n : {700, 1500}) add_msm(0, n, n, false);                                                       // engine slots
    for (size_t n : {4096, 6000}) add_msm(0, n, n, false);                                                      // split over the two devices
    add_msm(1, 3000, 9, false);
    for (size_t k : {5, 8, 16, 33, 64}) { add_pairing(k, true); add_pairing(k, false); }                        // coalesced
    add_pairing(200, true);                                                                                      // engine slots
    add_pairing(2500, true);                                                                                     // split
    add_pairing(2500, false);
    for (Case &c : cases) {
        c.want.assign(256, 0);
        c.rc = call(c, c.want.data());
    }
    int good_refs = 0;
    for (const Case &c : cases) {
        good_refs += c.rc == 0;
        if (c.rc) fprintf(stderr, "reference pass: kind %d, %zu bytes -> rc %d\n", c.kind, c.in.size(), c.rc);
    }
    if (good_refs < (int)cases.size() - 1) { fprintf(stderr, "reference pass: %d of %zu calls succeeded\n", good_refs, cases.size()); return 2; }

    std::atomic<long> bad{0}, done{0};
    std::atomic<bool> stop{false};
    const int nthreads = 24, rounds = 3;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            uint8_t out[256];
            for (int r = 0; r < rounds; r++)
                for (size_t i = 0; i < cases.size(); i++) {
                    const Case &c = cases[(i * 7 + (size_t)t * 3 + (size_t)r) % cases.size()];
                    memset(out, 0, sizeof out);
                    const int rc = call(c, out);
                    if (rc != c.rc || (rc == 0 && memcmp(out, c.want.data(), c.kind == 2 ? 32 : c.kind ? 256 : 128) != 0)) bad++;
                    done++;
                }
        });
    std::thread observer([&] {
        while (!stop.load()) {
            float p = 0, d = 0;
            char name[64];
            eip2537_hip_last_timing(&p, &d);
            (void)eip2537_hip_last_plan(name, sizeof name, nullptr, nullptr, nullptr, nullptr, nullptr);
            (void)eip2537_hip_trim(0);
            uint64_t x, y, z;
            eip2537_hip_coalesce_stats(&x, &y, &z);
            std::this_thread::yield();
        }
    });
    for (auto &t : th) t.join();
    stop = true;
    observer.join();
    uint64_t pipes = 0, calls = 0, largest = 0;
    eip2537_hip_coalesce_stats(&pipes, &calls, &largest);
    printf("hammer: %ld calls, %ld mismatches; coalesced %llu calls into %llu pipelines (largest batch %llu)\n", done.load(), bad.load(),
           (unsigned long long)calls, (unsigned long long)pipes, (unsigned long long)largest);
    return bad.load() ? 1 : 0;
}
