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

/* Devices.  Optional: every entry point initialises lazily.
 *   eip2537_hip_init(d), d >= 0   this process uses HIP device d only (one process per GPU: what
 *                                 bench.py's ranks do).  An ordinal that is not visible is an error.
 *   otherwise                     $EIP2537_HIP_DEVICES = "all" | comma list of ordinals, else
 *                                 $EIP2537_HIP_DEVICE = one ordinal, else every visible device.
 * With more than one device listed, the reference-ABI calls (bls12_g1multiexp / bls12_g2multiexp /
 * bls12_pairing, host input) cut a large input into contiguous record ranges, one per device and
 * host thread -- each device copies and reduces its own range -- and combine the partials on the
 * host (reference src/eip2537.c:541-561 has one CPU loop in that place); small inputs go to the
 * least busy device.  Returns 0, or EIP2537_MEMORY_ERROR when no such device exists or a different
 * selection was already made.  The caller's current HIP device is preserved across every call. */
int eip2537_hip_init(int device);
/* Number of devices host-input calls are spread over (initialises the selection); 0 = no device. */
int eip2537_hip_device_count(void);

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

/* Device time of the last GPU call made by this thread (else by the process), in milliseconds,
 * from HIP events on the engine's own stream: the whole device pipeline, and its dominant kernel
 * (the bucket accumulate of an MSM, the line walk of a pairing batch -- named by
 * eip2537_hip_last_plan).  For a call split over several devices: the slowest shard. */
void eip2537_hip_last_timing(float *pipeline_ms, float *dominant_kernel_ms);
/* Two more intervals of the same call, from HIP events on the engine's streams.  Pairing: aux1 = the G1 membership kernel
 * (second stream, beside the line walk), aux2 = the per-step line products (k_pair_fold + k_pair_tree2).  MSM: aux1 = the
 * sort stage (decode, histograms, scans, scatter, task order), aux2 = fold + bucket reduce. */
void eip2537_hip_last_timing_aux(float *aux1_ms, float *aux2_ms);
/* What that call ran: the dominant kernel's name as rocprofv3 prints it, the Pippenger window width
 * and window count (pairing: 0 and the 68 Miller steps), lanes per task / pair, the records / pairs
 * of the launch and the bucket count.  Any pointer may be NULL.  Returns 0, or EIP2537_EMPTY_INPUT
 * when no GPU call has been made yet. */
int eip2537_hip_last_plan(char *kernel_name, size_t cap, int *window_bits, int *windows, int *lanes,
                          uint32_t *units, uint32_t *buckets);
/* Start-up hook, optional: call ONCE from main(), before the process starts threads and before its first HIP call.  Sets
 * GPU_MAX_HW_QUEUES=16 unless the embedder has set it (the HIP runtime reads it when it initialises; with the default of 4,
 * concurrent callers' streams share 4 hardware queues).  The library never touches the environment from inside a precompile
 * call.  The static shim (shim/libblst_eip2537.a) exports the same name and also loads the engine.  Returns 0.
 * eip2537_hip_hw_queues(): the value the HIP runtime sees (4 when unset), so that an embedder can assert it. */
int eip2537_hip_early_init(void);
int eip2537_hip_hw_queues(void);
/* Record shards that call was staged in: a large host-input bls12_g1multiexp is copied shard by shard, each shard decoded,
 * sorted and accumulated behind its own copy into ONE bucket space (1: one copy, or input already in HBM; 0: no call yet). */
int eip2537_hip_last_shards(void);
/* Engine slots keep the workspace of the largest call they served (about 0.9 GB after one
 * 2^20-record MSM; a slot above $EIP2537_HIP_KEEP_MB, default 4096, frees it when the call ends).
 * This releases the workspace of every idle slot holding more than keep_bytes; returns the bytes freed. */
size_t eip2537_hip_trim(size_t keep_bytes);

/* Synthetic workloads (host code, for benchmarks and tests; not a precompile): records
 * i in [start, start+n) with P_i = [a + i*b]G and k_i = SplitMix64(seed) words 4i..4i+3;
 * pairing: pairs ([a0 + i*a1]G1, [b0 + i*b1]G2).  a/b are 32-byte little-endian integers. */
int eip2537_hip_gen_g1_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start);
int eip2537_hip_gen_g2_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t seed, uint64_t start);
int eip2537_hip_gen_pairing_input(uint8_t *out, size_t k, const uint8_t a0[32], const uint8_t a1[32],
                                  const uint8_t b0[32], const uint8_t b1[32], uint64_t start);

/* Concurrent small calls (multiexp above the host crossover and up to 512 records, pairing checks of 3..64 pairs, host input) are
 * coalesced: callers that arrive while the engine is busy are served together by ONE device pipeline over
 * their concatenated records, each with its own result and error code ($EIP2537_HIP_COALESCE=0 disables).  Counters since load: device
 * pipelines run for such calls, calls served, and the largest number of calls in one pipeline. */
void eip2537_hip_coalesce_stats(uint64_t *pipelines, uint64_t *calls, uint64_t *largest_batch);

/* Testing hook for the small-call crossover: the reference-ABI multiexp / pairing calls run the
 * library's own host code up to a measured size (G1 MSM 16 records, G2 MSM 6 -- the reference itself
 * forwards n == 1 to the mul precompile, src/eip2537.c:550-552 --, pairing 2 pairs:
 * profiles/r03_small_calls.txt) and the GPU above it.  route = 0: always the GPU; 1: the host code up to 64 units; -1: the default rule. */
int eip2537_hip_set_route(int route);

/* Testing hook: force the Pippenger window width (4..16); 0 restores the cost model. */
int eip2537_hip_set_window(int c);

/* Device self-test of the Fp products (column product and square, canonical and lazy forms) on n
 * pseudo-random and extreme operand pairs against an independent 12 x 32-bit product, all on the device.
 * mismatches[0..3] = product, square, lazy product, lazy square.  Returns 0 when the test ran. */
int eip2537_hip_field_selftest(uint64_t seed, size_t n, uint64_t mismatches[4]);
/* The same for the limb-form primitives the G1 MSM and every pairing kernel compute on (13 limbs of 30 bits, Montgomery
 * factor 2^390: products, squares, two-product sums with one reduction, weak reduction, linear steps, zero test), with
 * operands grown to the bounds the kernels use.  mismatches[0..7] = mulL, sqrL, mul2L, fp_mul2_cols30, limb conversion,
 * weak_reduceL / shlL, sums and differences, is_zero_modp. */
int eip2537_hip_limb_selftest(uint64_t seed, size_t n, uint64_t mismatches[8]);

#ifdef __cplusplus
}
#endif
#endif /* EIP2537_HIP_H_ */
