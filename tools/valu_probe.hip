// K0: issue-rate probe for the integer/FP64 multiply forms a 381-bit Montgomery product can be
// built from on gfx950.  Prints giga lane-ops/s for the whole chip (every CU busy, 8 waves/SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define ITERS 4096
#define CHAINS 8

template <int KIND>
__global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3u + blockIdx.x;
    uint64_t acc[CHAINS];
    double d[CHAINS];
    uint32_t u[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { acc[i] = a + i; d[i] = (double)(a + i); u[i] = a * (i + 1); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (KIND == 0) acc[i] = (uint64_t)(uint32_t)acc[i] * b + acc[i];            // v_mad_u64_u32
            if (KIND == 1) u[i] = u[i] * b + a;                                          // v_mul_lo_u32 + add (or mad_u32)
            if (KIND == 2) u[i] = __umulhi(u[i], b) + a;                                 // v_mul_hi_u32
            if (KIND == 3) u[i] = __umul24(u[i], b) + a;                                // v_mad_u32_u24
            if (KIND == 4) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);                   // v_fma_f64
            if (KIND == 5) u[i] = u[i] + b + (u[i] >> 3);                                // plain 32-bit adds
            // the carry handling around the multiply-adds of the limb form (round 3): is a 64-bit shift / add full rate?
            if (KIND == 6) asm volatile("v_lshrrev_b64 %0, 3, %1" : "=v"(acc[i]) : "v"(acc[i]));
            if (KIND == 7) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(acc[i]) : "v"(acc[i]), "v"(acc[(i + 1) % CHAINS]));
            if (KIND == 8) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(b));
            if (KIND == 9) asm volatile("v_alignbit_b32 %0, %1, %2, 30" : "=v"(u[i]) : "v"(u[i]), "v"(b));
            if (KIND == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %2" : "=v"(acc[i]) : "v"(u[i]), "v"(acc[i]) : "vcc");
            if (KIND == 11) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(b));
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ u[i] ^ (uint32_t)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND> static void run(const char *name, uint32_t *d_out, double ops_per_iter) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<KIND><<<blocks, 256>>>(d_out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) probe<KIND><<<blocks, 256>>>(d_out, 777u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = 5.0 * blocks * 256.0 * ITERS * CHAINS * ops_per_iter;
    printf("%-28s %8.1f G lane-ops/s  (%.3f ms)\n", name, lane_ops / (ms * 1e-3) / 1e9, ms / 5);
}

// Sustained form (round 4): back-to-back v_mad_u64_u32 launches for `seconds`, the rate printed per ~0.25 s window -- the burst
// figure above is a 0.6 ms probe, and the part lowers its clock under sustained VALU load.
static void sustained(uint32_t *d_out, double seconds) {
    const int blocks = 256 * 8, per = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double elapsed = 0, first = 0, last = 0, sum = 0;
    int windows = 0;
    while (elapsed < seconds) {
        hipEventRecord(e0);
        for (int r = 0; r < per; r++) probe<0><<<blocks, 256>>>(d_out, 777u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double rate = (double)per * blocks * 256.0 * ITERS * CHAINS / (ms * 1e-3) / 1e9;
        printf("  t = %6.3f s   %8.1f G lane-ops/s\n", elapsed, rate);
        if (!windows) first = rate;
        last = rate;
        sum += rate;
        windows++;
        elapsed += ms * 1e-3;
    }
    printf("v_mad_u64_u32 sustained %.2f s: first window %.1f, last %.1f, mean %.1f G lane-ops/s\n", elapsed, first, last, sum / windows);
}

int main(int argc, char **argv) {
    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    if (argc > 1) { sustained(d_out, atof(argv[1])); return 0; }
    run<0>("v_mad_u64_u32", d_out, 1);
    run<1>("v_mul_lo_u32(+add)", d_out, 1);
    run<2>("v_mul_hi_u32(+add)", d_out, 1);
    run<3>("v_mad_u32_u24", d_out, 1);
    run<4>("v_fma_f64", d_out, 1);
    run<5>("v_add_u32 x2 + shift", d_out, 3);
    run<6>("v_lshrrev_b64", d_out, 1);
    run<7>("v_lshl_add_u64", d_out, 1);
    run<8>("v_and_b32", d_out, 1);
    run<9>("v_alignbit_b32", d_out, 1);
    run<10>("v_mad_u64_u32 (x, 1, acc)", d_out, 1);
    run<11>("v_mul_lo_u32", d_out, 1);
    return 0;
}
