// Batched pairing check on gfx950: one (G1, G2) pair per lane.
//
// Replaces the reference's sequential loop (src/eip2537.c:1033-1068): per pair decode G1, G1
// subgroup test, decode G2, G2 subgroup test, Miller loop, running Fp12 product; then ONE final
// exponentiation for the whole batch (:1070) and the == 1 test (:1076).
//
// Three independent kernels run concurrently on three streams:
//   k_pair_check_g1  [pair]  decode + on-curve + G1 membership  (phi(P) == -[z^2]P)
//   k_pair_check_g2  [pair]  decode + on-curve + G2 membership  (psi(Q) == [z]Q)
//   k_pair_miller    [pair]  Miller loop f_i, then a per-wave Fp12 product tree over shuffles
// Errors are merged with atomicMin on (pair << 4 | stage << 3 | code): lowest pair first, and
// inside a pair the reference's order G1 decode -> G1 subgroup -> G2 decode -> G2 subgroup.
// The host multiplies the per-wave products and runs the single final exponentiation.
#include <stdio.h>
#include <vector>
#include "codec.h"
#include "pairing.h"
#include "engine.h"

namespace eip {

#define HIPCHK(x)                                                                               \
    do {                                                                                        \
        hipError_t _e = (x);                                                                    \
        if (_e != hipSuccess) {                                                                 \
            fprintf(stderr, "[eip2537_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), \
                    __FILE__, __LINE__);                                                        \
            return E_MEMORY_ERROR;                                                              \
        }                                                                                       \
    } while (0)

static constexpr int kPairWords = 96;   // 384 bytes

__global__ void __launch_bounds__(64)
k_pair_check_g1(const uint32_t *__restrict__ in, uint32_t k, unsigned long long *err) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= k) return;
    Aff<Fp> p;
    int st = decode_point<Fp>(p, in + (size_t)i * kPairWords);
    if (st == E_SUCCESS && !in_g1(p)) st = E_NOT_IN_SUBGROUP;
    if (st != E_SUCCESS) atomicMin(err, ((unsigned long long)i << 4) | (unsigned long long)st);
}

__global__ void __launch_bounds__(64)
k_pair_check_g2(const uint32_t *__restrict__ in, uint32_t k, unsigned long long *err) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= k) return;
    Aff<Fp2> q;
    int st = decode_point<Fp2>(q, in + (size_t)i * kPairWords + 32);
    if (st == E_SUCCESS && !in_g2(q)) st = E_NOT_IN_SUBGROUP;
    if (st != E_SUCCESS) atomicMin(err, ((unsigned long long)i << 4) | 8ull | (unsigned long long)st);
}

__device__ __forceinline__ Fp shfl_down(const Fp &a, int off) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = __shfl_down(a.l[i], off, 64);
    return r;
}
__device__ __forceinline__ Fp2 shfl_down(const Fp2 &a, int off) { return Fp2{shfl_down(a.c0, off), shfl_down(a.c1, off)}; }
__device__ __forceinline__ Fp6 shfl_down(const Fp6 &a, int off) { return Fp6{shfl_down(a.a0, off), shfl_down(a.a1, off), shfl_down(a.a2, off)}; }
__device__ __forceinline__ Fp12 shfl_down(const Fp12 &a, int off) { return Fp12{shfl_down(a.c0, off), shfl_down(a.c1, off)}; }

__global__ void __launch_bounds__(64)
k_pair_miller(const uint32_t *__restrict__ in, uint32_t k, Fp12 *__restrict__ wave_out) {
    uint32_t i = blockIdx.x * 64u + threadIdx.x;
    Fp12 f = fp12_one();
    if (i < k) {
        Aff<Fp> p;
        Aff<Fp2> q;
        int s1 = decode_point<Fp>(p, in + (size_t)i * kPairWords);
        int s2 = decode_point<Fp2>(q, in + (size_t)i * kPairWords + 32);
        if (s1 == E_SUCCESS && s2 == E_SUCCESS) f = miller_loop(p, q);   // else: result discarded by the caller
    }
    const int lane = threadIdx.x & 63;
    for (int off = 32; off >= 1; off >>= 1) {
        Fp12 o = shfl_down(f, off);
        if (lane < off) f = mul(f, o);
    }
    if (lane == 0) wave_out[blockIdx.x] = f;
}

int pairing_device(Engine *e, const void *d_in, size_t k, uint32_t *ml_words) {
    if (k == 0 || k >= (1ull << 31)) return E_MEMORY_ERROR;
    if ((reinterpret_cast<uintptr_t>(d_in) & 3u) != 0) {
        fprintf(stderr, "[eip2537_hip] device input must be 4-byte aligned\n");
        return E_MEMORY_ERROR;
    }
    const uint32_t blocks = (uint32_t)((k + 63) / 64);
    HIPCHK(e->misc.reserve(64));
    HIPCHK(e->winout.reserve((size_t)blocks * sizeof(Fp12)));
    auto *err = reinterpret_cast<unsigned long long *>(e->misc.p);
    auto *wave_out = reinterpret_cast<Fp12 *>(e->winout.p);
    const uint32_t *in = reinterpret_cast<const uint32_t *>(d_in);

    hipStream_t s = e->stream;
    HIPCHK(hipMemsetAsync(err, 0xFF, 8, s));
    HIPCHK(hipEventRecord(e->ev_start, s));
    // fork: the two membership kernels run beside the Miller loops
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_start, 0));
    HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_start, 0));
    hipLaunchKernelGGL(k_pair_check_g1, dim3(blocks), dim3(64), 0, e->stream2, in, (uint32_t)k, err);
    hipLaunchKernelGGL(k_pair_check_g2, dim3(blocks), dim3(64), 0, e->stream3, in, (uint32_t)k, err);
    HIPCHK(hipEventRecord(e->ev_a, s));
    hipLaunchKernelGGL(k_pair_miller, dim3(blocks), dim3(64), 0, s, in, (uint32_t)k, wave_out);
    HIPCHK(hipEventRecord(e->ev_b, s));
    HIPCHK(hipEventRecord(e->ev_j2, e->stream2));
    HIPCHK(hipEventRecord(e->ev_j3, e->stream3));
    HIPCHK(hipStreamWaitEvent(s, e->ev_j2, 0));
    HIPCHK(hipStreamWaitEvent(s, e->ev_j3, 0));
    HIPCHK(hipEventRecord(e->ev_stop, s));
    HIPCHK(hipGetLastError());

    unsigned long long herr = 0;
    std::vector<Fp12> parts(blocks);
    HIPCHK(hipMemcpyAsync(&herr, err, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(parts.data(), wave_out, (size_t)blocks * sizeof(Fp12), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev_start, e->ev_stop) == hipSuccess) e->last_kernel_ms = ms;
    if (hipEventElapsedTime(&ms, e->ev_a, e->ev_b) == hipSuccess) e->last_accum_ms = ms;
    if (herr != ~0ull) return (int)(herr & 7ull);

    Fp12 acc = parts[0];
    for (uint32_t b = 1; b < blocks; b++) acc = mul(acc, parts[b]);
    memcpy(ml_words, &acc, sizeof acc);
    return E_SUCCESS;
}

}  // namespace eip
