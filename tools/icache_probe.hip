// Does a latency-bound kernel whose hot loop is larger than the instruction cache run slower per instruction?
// One wave per SIMD (1024 blocks of 64 threads, each claiming its SIMD), a dependent chain of limb-form products
// (csrc/limb30.h, ~3.7 KB of code each) unrolled U times inside a rolled loop: the loop body is U x 3.7 KB.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iblst_eip2537_amd/csrc -Iinclude tools/icache_probe.hip -o icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "limb30.h"
using namespace eip;

template <int U, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_chain(const FpL *in, FpL *out, int iters) {
    asm volatile("v_accvgpr_write_b32 a127, 0" ::: "a127");
    FpL x = in[threadIdx.x & 63], y = in[64 + (threadIdx.x & 63)];
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            x = mulL(x, y);
            asm volatile("" : "+v"(x.l[0]));            // keep the copies apart (no common-subexpression games)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <int U, int WAVES> static void run(const FpL *d_in, FpL *d_out, int blocks) {
    const int total = 4096, iters = total / U;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_chain<U, WAVES><<<blocks, 64 * WAVES>>>(d_in, d_out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_chain<U, WAVES><<<blocks, 64 * WAVES>>>(d_in, d_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("unroll %3d (loop body ~%5.0f KB)  blocks %4d x %d waves: %7.3f ms  = %6.1f ns per product per wave\n", U, U * 3.7, blocks, WAVES, ms,
           ms * 1e6 / total);
}

int main() {
    FpL h[128];
    for (int i = 0; i < 128; i++) for (int k = 0; k < 13; k++) h[i].l[k] = (0x12345u * (i + 1) + 0x9e3779u * k) & (k < 12 ? 0x3fffffffu : 0xfffffu);
    FpL *d_in, *d_out;
    (void)hipMalloc(&d_in, sizeof h);
    (void)hipMalloc(&d_out, 4096 * 256 * sizeof(FpL));
    (void)hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    run<1, 1>(d_in, d_out, 1024);
    run<4, 1>(d_in, d_out, 1024);
    run<8, 1>(d_in, d_out, 1024);
    run<16, 1>(d_in, d_out, 1024);
    run<32, 1>(d_in, d_out, 1024);
    run<64, 1>(d_in, d_out, 1024);
    printf("four waves per block (one per SIMD of a CU), 256 blocks:\n");
    run<4, 4>(d_in, d_out, 256);
    run<32, 4>(d_in, d_out, 256);
    run<64, 4>(d_in, d_out, 256);
    printf("one wave on the whole chip:\n");
    run<4, 1>(d_in, d_out, 1);
    run<64, 1>(d_in, d_out, 1);
    return 0;
}
