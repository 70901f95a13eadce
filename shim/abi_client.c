/* A C caller shaped like the reference's FFI users (rust/src/lib.rs:18-96: (*mut u8, *const u8,
 * usize) -> u32): links ONLY the static shim + libc, calls through it from two threads, and
 * makes the zero-length call with a dangling non-null pointer that Rust makes.  Used by
 * tests/test_shim.py. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../include/eip2537.h"

static const unsigned char G1[128] = {
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
    0x17,0xf1,0xd3,0xa7,0x31,0x97,0xd7,0x94,0x26,0x95,0x63,0x8c,0x4f,0xa9,0xac,0x0f,0xc3,0x68,0x8c,0x4f,0x97,0x74,0xb9,0x05,
    0xa1,0x4e,0x3a,0x3f,0x17,0x1b,0xac,0x58,0x6c,0x55,0xe8,0x3f,0xf9,0x7a,0x1a,0xef,0xfb,0x3a,0xf0,0x0a,0xdb,0x22,0xc6,0xbb,
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
    0x08,0xb3,0xf4,0x81,0xe3,0xaa,0xa0,0xf1,0xa0,0x9e,0x30,0xed,0x74,0x1d,0x8a,0xe4,0xfc,0xf5,0xe0,0x95,0xd5,0xd0,0x0a,0xf6,
    0x00,0xdb,0x18,0xcb,0x2c,0x04,0xb3,0xed,0xd0,0x3c,0xc7,0x44,0xa2,0x88,0x8a,0xe4,0x0c,0xaa,0x23,0x29,0x46,0xc5,0xe7,0xe1};

static void *worker(void *arg) {
    long bad = 0;
    unsigned char in[256], sum[128], dbl[128], mul_in[160];
    memcpy(in, G1, 128);
    memcpy(in + 128, G1, 128);
    memcpy(mul_in, G1, 128);
    memset(mul_in + 128, 0, 32);
    mul_in[159] = 2;
    for (int i = 0; i < 50; i++) {
        if (bls12_g1add(sum, in, 256) != EIP2537_SUCCESS) bad++;
        if (bls12_g1mul(dbl, mul_in, 160) != EIP2537_SUCCESS) bad++;
        if (memcmp(sum, dbl, 128)) bad++;                                   /* G + G == [2]G */
        if (bls12_g1add(sum, in, 255) != EIP2537_INVALID_LENGTH) bad++;
        if (bls12_g1multiexp(sum, (byte *)(uintptr_t)1, 0) != EIP2537_INVALID_LENGTH) bad++;   /* dangling, len 0 */
    }
    *(long *)arg = bad;
    return NULL;
}

int main(void) {
    pthread_t t[2];
    long bad[2] = {0, 0};
    for (int i = 0; i < 2; i++) pthread_create(&t[i], NULL, worker, &bad[i]);
    for (int i = 0; i < 2; i++) pthread_join(t[i], NULL);
    if (bls12_g1multiexp_gas(160 * 2) != 2ull * 12000 * 888 / 1000 || BLS12_G1ADD_GAS != 600) bad[0]++;
    printf("shim client: %ld failures\n", bad[0] + bad[1]);
    return (bad[0] + bad[1]) ? 1 : 0;
}
