/*
 * eip2537_hip.h -- extension entry points of libeip2537_hip.so beyond the reference ABI.
 *
 * The reference has no device or multi-GPU notion (single-threaded CPU code); these functions
 * exist so that (a) a caller that already holds the encoded input in HBM can skip the host
 * copy (bench.py's timed region) and (b) one MSM / pairing batch can be sharded
 * over several GPUs, one process per GPU, with the partial results exchanged as plain bytes
 * (SURVEY.md 8e).  Same conventions as eip2537.h: plain pointers and sizes, return code 0 =
 * success, otherwise an EIP2537_ERROR value; outputs written only on success.
 *
 *   n_records / n_pairs count 160-byte (G1 MSM), 288-byte (G2 MSM) or 384-byte (pairing) records
 *   laid out exactly as the reference's bls12_g1multiexp / bls12_g2multiexp / bls12_pairing input
 *   (reference src/eip2537.c:541-548, 829-836, 1020-1027).  d_in is a device pointer, 4-byte aligned.
 */
#ifndef EIP2537_HIP_H_
#define EIP2537_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIP2537_HIP_G1_PARTIAL_BYTES 192   /* XYZZ point over Fp, Montgomery limbs  */
#define EIP2537_HIP_G2_PARTIAL_BYTES 384   /* XYZZ point over Fp2                   */
#define EIP2537_HIP_ML_PARTIAL_BYTES 576   /* Fp12 Miller-loop product              */

/* Select the HIP device for this process (default: device 0, or $EIP2537_HIP_DEVICE) and
 * initialise the engine.  Optional: every entry point initialises lazily. */
int eip2537_hip_init(int device);

/* Full precompile on an input already resident in HBM. */
int eip2537_hip_g1multiexp_dev(uint8_t out[128], const void *d_in, size_t n_records);
int eip2537_hip_g2multiexp_dev(uint8_t out[256], const void *d_in, size_t n_records);
int eip2537_hip_pairing_dev(uint8_t out[32], const void *d_in, size_t n_pairs);

/* Sharded form: each rank reduces its contiguous record range to one partial ... */
int eip2537_hip_g1msm_partial_dev(uint8_t partial[EIP2537_HIP_G1_PARTIAL_BYTES], const void *d_in, size_t n_records);
int eip2537_hip_g2msm_partial_dev(uint8_t partial[EIP2537_HIP_G2_PARTIAL_BYTES], const void *d_in, size_t n_records);
int eip2537_hip_pairing_partial_dev(uint8_t partial[EIP2537_HIP_ML_PARTIAL_BYTES], const void *d_in, size_t n_pairs);
/* ... and any rank combines `count` gathered partials into the precompile's output. */
int eip2537_hip_g1msm_combine(uint8_t out[128], const uint8_t *partials, size_t count);
int eip2537_hip_g2msm_combine(uint8_t out[256], const uint8_t *partials, size_t count);
int eip2537_hip_pairing_combine(uint8_t out[32], const uint8_t *partials, size_t count);

/* Device time of the last GPU call made by this process, in milliseconds, from HIP events on
 * the engine's own stream: the whole device pipeline, and its dominant kernel (k_msm_accum for
 * an MSM, k_pair_miller for a pairing batch). */
void eip2537_hip_last_timing(float *pipeline_ms, float *dominant_kernel_ms);

/* Synthetic workloads (host code, for benchmarks and tests; not a precompile): records
 * i in [start, start+n) with P_i = [a + i*b]G and k_i = SplitMix64(seed) words 4i..4i+3;
 * pairing: pairs ([a0 + i*a1]G1, [b0 + i*b1]G2).  a/b are 32-byte little-endian integers. */
int eip2537_hip_gen_g1_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start);
int eip2537_hip_gen_g2_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start);
int eip2537_hip_gen_pairing_input(uint8_t *out, size_t k, const uint8_t a0[32], const uint8_t a1[32],
                                  const uint8_t b0[32], const uint8_t b1[32], uint64_t start);

/* Testing hook: force the Pippenger window width (4..16); 0 restores the cost model. */
int eip2537_hip_set_window(int c);

#ifdef __cplusplus
}
#endif
#endif /* EIP2537_HIP_H_ */
