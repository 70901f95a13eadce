// H2D probe (round 4): how fast does a pageable 168 MB buffer reach HBM -- one hipMemcpyAsync, or two threads copying halves on two streams?
//   hipcc -O2 --offload-arch=gfx950 tools/h2d_probe.hip -lpthread -o /tmp/h2d_probe && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = (size_t)160 << 20;
    char *h = (char *)malloc(N);
    memset(h, 1, N);
    char *d;
    if (hipMalloc(&d, N) != hipSuccess) return 1;
    hipStream_t s[4];
    for (auto &x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    auto run = [&](int threads, const char *name) {
        double best = 1e9, sum = 0;
        const int reps = 8;
        for (int r = 0; r < reps + 2; r++) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 1; t < threads; t++)
                th.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    const size_t lo = N * t / threads, hi = N * (t + 1) / threads;
                    (void)hipMemcpyAsync(d + lo, h + lo, hi - lo, hipMemcpyHostToDevice, s[t]);
                    (void)hipStreamSynchronize(s[t]);
                });
            (void)hipMemcpyAsync(d, h, N / threads, hipMemcpyHostToDevice, s[0]);
            (void)hipStreamSynchronize(s[0]);
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            if (r >= 2) { best = dt < best ? dt : best; sum += dt; }
        }
        printf("%-44s best %.3f ms (%.1f GB/s)  mean %.3f ms\n", name, best * 1e3, N / best / 1e9, sum / reps * 1e3);
    };
    run(1, "pageable, 1 thread");
    run(2, "pageable, 2 threads x 2 streams (halves)");
    run(3, "pageable, 3 threads x 3 streams");
    run(4, "pageable, 4 threads x 4 streams");
    {   // registration cost and the pinned rate, for reference
        const double t0 = now();
        const hipError_t e = hipHostRegister(h, N, hipHostRegisterDefault);
        const double t1 = now();
        printf("hipHostRegister(160 MiB): %s, %.3f ms\n", e == hipSuccess ? "ok" : "failed", (t1 - t0) * 1e3);
        if (e == hipSuccess) {
            run(1, "registered (pinned), 1 thread");
            const double t2 = now();
            (void)hipHostUnregister(h);
            printf("hipHostUnregister: %.3f ms\n", (now() - t2) * 1e3);
        }
    }
    return 0;
}
