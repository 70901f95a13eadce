/*
 * libblst_eip2537.a -- pure-C forwarding shim for the reference's consumers.
 *
 * The reference's Rust build script links a prebuilt `rust/libblst_eip2537.a` when it exists and
 * adds no other -l flags (reference rust/build.rs:37-41); its Go package compiles `src/eip2537.c`
 * by #include (reference go/blst_eip2537_c_files.c:7) with no LDFLAGS.  Neither can therefore
 * depend on libamdhip64 / libstdc++ at link time.  This file is that archive's only member (and
 * can be dropped in as `src/eip2537.c` for cgo): it exports the reference ABI
 * (reference src/eip2537.h:42-82), depends on libc only, and on first use dlopen()s the real engine,
 * libeip2537_hip.so ($EIP2537_HIP_LIB, else the default search path), then forwards every call.
 * If the engine cannot be loaded every call fails loudly with EIP2537_MEMORY_ERROR.
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned char byte;
enum { SHIM_MEMORY_ERROR = 7 };

typedef int (*precompile_fn)(byte *, const byte *, size_t);

/* The engine handle and every resolved entry point are published with release / acquire atomics:
 * callers are concurrent (cargo test threads, goroutines) and the first calls race.  Two threads may
 * both dlopen / dlsym -- the loader reference-counts the handle and both get the same addresses --
 * after which every call is one atomic load and an indirect call (no dlsym on the hot path). */
static void *g_handle;

static void *shim_handle(void) {
    void *h = __atomic_load_n(&g_handle, __ATOMIC_ACQUIRE);
    if (h) return h;
    const char *path = getenv("EIP2537_HIP_LIB");
    h = dlopen(path && *path ? path : "libeip2537_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "[eip2537 shim] cannot load the HIP engine: %s (set EIP2537_HIP_LIB)\n", dlerror());
        return NULL;
    }
    __atomic_store_n(&g_handle, h, __ATOMIC_RELEASE);
    return h;
}

/* Optional start-up hook (include/eip2537_hip.h): call it ONCE from main(), before the process starts threads and before its
 * first HIP call.  The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (default 4: concurrent callers would share 4
 * hardware queues); a lazy dlopen from a caller thread cannot set it safely (setenv races with getenv elsewhere), this can.
 * The embedder's own value is never overridden.  Also loads the engine, so that the first precompile call does not pay for it. */
int eip2537_hip_early_init(void) {
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    return shim_handle() ? 0 : SHIM_MEMORY_ERROR;
}

static int shim_forward(precompile_fn *slot, const char *name, byte *out, const byte *in, size_t len) {
    precompile_fn f = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
    if (!f) {
        void *h = shim_handle();
        if (!h) return SHIM_MEMORY_ERROR;
        f = (precompile_fn)dlsym(h, name);
        if (!f) {
            fprintf(stderr, "[eip2537 shim] engine lacks symbol %s\n", name);
            return SHIM_MEMORY_ERROR;
        }
        __atomic_store_n(slot, f, __ATOMIC_RELEASE);
    }
    return f(out, in, len);
}

#define FORWARD(name)                                                                         \
    static precompile_fn shim_fn_##name;                                                      \
    int name(byte *out, const byte *in, size_t in_len) { return shim_forward(&shim_fn_##name, #name, out, in, in_len); }

FORWARD(bls12_g1add)
FORWARD(bls12_g1mul)
FORWARD(bls12_g1multiexp)
FORWARD(bls12_g1multiexp_naive)
FORWARD(bls12_g1multiexp_bc)
FORWARD(bls12_g2add)
FORWARD(bls12_g2mul)
FORWARD(bls12_g2multiexp)
FORWARD(bls12_g2multiexp_naive)
FORWARD(bls12_g2multiexp_bc)
FORWARD(bls12_pairing)
FORWARD(bls12_map_fp_to_g1)
FORWARD(bls12_map_fp2_to_g2)

/* gas schedule: data symbols cannot be forwarded, so the table lives here too
 * (values: reference src/eip2537.c:1169-1197) */
const uint64_t BLS12_G1ADD_GAS = 600;
const uint64_t BLS12_G1MUL_GAS = 12000;
const uint64_t BLS12_G2ADD_GAS = 4500;
const uint64_t BLS12_G2MUL_GAS = 55000;
const uint64_t BLS12_PAIRING_BASE_GAS = 115000;
const uint64_t BLS12_PAIRING_PAIR_GAS = 23000;
const uint64_t BLS12_MAP_FP_TO_G1_GAS = 5500;
const uint64_t BLS12_MAP_FP2_TO_G2_GAS = 110000;
const uint64_t BLS12_MULTIEXP_MULTIPLIER_GAS = 1000;
const uint64_t BLS12_MULTIEXP_DISCOUNT_TABLE_LEN = 128;
const uint64_t BLS12_MULTIEXP_DISCOUNT[128] = {
    1200, 888, 764, 641, 594, 547, 500, 453, 438, 423, 408, 394, 379, 364, 349, 334,
    330, 326, 322, 318, 314, 310, 306, 302, 298, 294, 289, 285, 281, 277, 273, 269,
    268, 266, 265, 263, 262, 260, 259, 257, 256, 254, 253, 251, 250, 248, 247, 245,
    244, 242, 241, 239, 238, 236, 235, 233, 232, 231, 229, 228, 226, 225, 223, 222,
    221, 220, 219, 219, 218, 217, 216, 216, 215, 214, 213, 213, 212, 211, 211, 210,
    209, 208, 208, 207, 206, 205, 205, 204, 203, 202, 202, 201, 200, 199, 199, 198,
    197, 196, 196, 195, 194, 193, 193, 192, 191, 191, 190, 189, 188, 188, 187, 186,
    185, 185, 184, 183, 182, 182, 181, 180, 179, 179, 178, 177, 176, 176, 175, 174};

static uint64_t shim_msm_gas(uint64_t len, uint64_t rec, uint64_t mul_gas) {
    uint64_t k = len / rec;
    if (k == 0) return 0;
    return k * mul_gas * BLS12_MULTIEXP_DISCOUNT[k < 128 ? k - 1 : 127] / 1000;
}
uint64_t bls12_g1add_gas(void) { return BLS12_G1ADD_GAS; }
uint64_t bls12_g1mul_gas(void) { return BLS12_G1MUL_GAS; }
uint64_t bls12_g1multiexp_gas(uint64_t len) { return shim_msm_gas(len, 160, BLS12_G1MUL_GAS); }
uint64_t bls12_g2add_gas(void) { return BLS12_G2ADD_GAS; }
uint64_t bls12_g2mul_gas(void) { return BLS12_G2MUL_GAS; }
uint64_t bls12_g2multiexp_gas(uint64_t len) { return shim_msm_gas(len, 288, BLS12_G2MUL_GAS); }
uint64_t bls12_pairing_gas(uint64_t len) { uint64_t k = len / 384; return k ? 115000 + 23000 * k : 0; }
uint64_t bls12_map_fp_to_g1_gas(void) { return BLS12_MAP_FP_TO_G1_GAS; }
uint64_t bls12_map_fp2_to_g2_gas(void) { return BLS12_MAP_FP2_TO_G2_GAS; }
