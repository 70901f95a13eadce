/*
 * eip2537.h -- C-ABI of the MI355X-native EIP-2537 engine (libeip2537_hip.so).
 *
 * Drop-in for the reference's src/eip2537.h: the same 13 entry points, the same error enum
 * values, the same gas symbols, so the reference's Rust `extern "C"` block
 * (rust/src/lib.rs:18-96) and Go cgo wrappers (go/blst_eip2537.go:44-211) bind unchanged.
 * The only difference is that this header does not include blst.h: `byte` is defined here.
 *
 *   reference interface replaced                       this header
 *   -------------------------------------------------  -----------------------------
 *   EIP2537_ERROR                src/eip2537.h:31-40   EIP2537_ERROR (same values)
 *   bls12_g1add / g1mul          src/eip2537.h:42-43   host path of the engine
 *   bls12_g1multiexp(_naive,_bc) src/eip2537.h:44-46   GPU Pippenger (all three names)
 *   bls12_g2add / g2mul          src/eip2537.h:48-49   host path of the engine
 *   bls12_g2multiexp(_naive,_bc) src/eip2537.h:50-52   GPU Pippenger (all three names)
 *   bls12_pairing                src/eip2537.h:54      GPU batched Miller loops
 *   bls12_map_fp_to_g1/_fp2_to_g2 src/eip2537.h:56-59  host path of the engine (RFC 9380 SSWU)
 *   gas constants / functions    src/eip2537.h:62-82   identical values
 *
 * Conventions (reference src/eip2537.c): `out` is caller-allocated and written only on success;
 * `in` is never written and never retained after return; in_len rules and the error code of the
 * lowest-index bad record are those of the reference.
 */
#ifndef EIP2537_H_
#define EIP2537_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __BLST_H__
typedef unsigned char byte;
#endif

typedef enum {
  EIP2537_SUCCESS = 0,
  EIP2537_POINT_NOT_ON_CURVE,
  EIP2537_POINT_NOT_IN_SUBGROUP,
  EIP2537_INVALID_ELEMENT,
  EIP2537_ENCODING_ERROR,
  EIP2537_INVALID_LENGTH,
  EIP2537_EMPTY_INPUT,
  EIP2537_MEMORY_ERROR,
} EIP2537_ERROR;

EIP2537_ERROR bls12_g1add(byte out[128], const byte in[256], size_t in_len);
EIP2537_ERROR bls12_g1mul(byte out[128], const byte in[160], size_t in_len);
EIP2537_ERROR bls12_g1multiexp(byte out[128], byte *in, size_t in_len);
EIP2537_ERROR bls12_g1multiexp_naive(byte out[128], byte *in, size_t in_len);
EIP2537_ERROR bls12_g1multiexp_bc(byte out[128], byte *in, size_t in_len);

EIP2537_ERROR bls12_g2add(byte out[256], const byte in[512], size_t in_len);
EIP2537_ERROR bls12_g2mul(byte out[256], const byte in[288], size_t in_len);
EIP2537_ERROR bls12_g2multiexp(byte out[256], byte *in, size_t in_len);
EIP2537_ERROR bls12_g2multiexp_naive(byte out[256], byte *in, size_t in_len);
EIP2537_ERROR bls12_g2multiexp_bc(byte out[256], byte *in, size_t in_len);

EIP2537_ERROR bls12_pairing(byte out[32], byte *in, size_t in_len);

EIP2537_ERROR bls12_map_fp_to_g1(byte out[128], const byte in[64], size_t in_len);
EIP2537_ERROR bls12_map_fp2_to_g2(byte out[256], const byte in[128], size_t in_len);

extern const uint64_t BLS12_G1ADD_GAS;
extern const uint64_t BLS12_G1MUL_GAS;
extern const uint64_t BLS12_G2ADD_GAS;
extern const uint64_t BLS12_G2MUL_GAS;
extern const uint64_t BLS12_PAIRING_BASE_GAS;
extern const uint64_t BLS12_PAIRING_PAIR_GAS;
extern const uint64_t BLS12_MAP_FP_TO_G1_GAS;
extern const uint64_t BLS12_MAP_FP2_TO_G2_GAS;
extern const uint64_t BLS12_MULTIEXP_MULTIPLIER_GAS;
extern const uint64_t BLS12_MULTIEXP_DISCOUNT_TABLE_LEN;
extern const uint64_t BLS12_MULTIEXP_DISCOUNT[128];

uint64_t bls12_g1add_gas(void);
uint64_t bls12_g1mul_gas(void);
uint64_t bls12_g1multiexp_gas(uint64_t input_len);
uint64_t bls12_g2add_gas(void);
uint64_t bls12_g2mul_gas(void);
uint64_t bls12_g2multiexp_gas(uint64_t input_len);
uint64_t bls12_pairing_gas(uint64_t input_len);
uint64_t bls12_map_fp_to_g1_gas(void);
uint64_t bls12_map_fp2_to_g2_gas(void);

#ifdef __cplusplus
}
#endif
#endif /* EIP2537_H_ */
