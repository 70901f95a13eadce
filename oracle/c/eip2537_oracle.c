/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's precompile logic
 * (sean-sn/blst_eip2537, src/eip2537.c) on the from-scratch arithmetic in ora_field.h / ora_ec.h.
 *
 * PARITY UNPINNED: the arithmetic library the reference calls (supranational/blst, unpinned,
 * cloned by build.sh:3-5) is not under /root/reference and the reference's known-answer files
 * (build.sh:13-52) are absent, so no golden vector of the reference pins these outputs.  What
 * pins them instead is listed in DESIGN.md ("Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Exported symbols are prefixed oracle_ so they can never satisfy the product's bls12_* ABI.
 *
 * Control flow restated (file:line of the reference):
 *   fp_from_bytes / fp_to_bytes                  src/eip2537.c:263-317
 *   decode/encode G1, Fp2, G2, scalar            src/eip2537.c:320-420
 *   bls12_g1add / g1mul                          src/eip2537.c:434-524
 *   bls12_g1multiexp dispatcher (1 / <=4 / else) src/eip2537.c:541-561
 *   bls12_g1multiexp_naive                       src/eip2537.c:564-616
 *   Bos-Coster heap + driver                     src/eip2537.c:57-205, 619-708
 *   G2 clones                                    src/eip2537.c:208-259, 722-998
 *   bls12_pairing                                src/eip2537.c:1020-1081
 *   bls12_map_fp_to_g1 / bls12_map_fp2_to_g2     src/eip2537.c:1094-1165 (RFC 9380 SSWU + isogeny,
 *                                                pinned by the RFC's appendix-J vectors)
 *   gas schedule                                 src/eip2537.c:1168-1271
 */
#include <stdlib.h>
#include "ora_ec.h"
#include "ora_iso_constants.h"

enum {
    ORA_SUCCESS = 0, ORA_POINT_NOT_ON_CURVE, ORA_POINT_NOT_IN_SUBGROUP, ORA_INVALID_ELEMENT,
    ORA_ENCODING_ERROR, ORA_INVALID_LENGTH, ORA_EMPTY_INPUT, ORA_MEMORY_ERROR
};
#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ codec */
/* returns -1 invalid, 0 zero, 1 non-zero (src/eip2537.c:263-309) */
static int ora_fp_from_bytes(fp *out, const uint8_t *in) {
    uint8_t pad = 0;
    for (int i = 0; i < 16; i++) pad |= in[i];
    if (pad) return -1;
    fp raw;
    for (int limb = 0; limb < 6; limb++) {
        uint64_t w = 0;
        const uint8_t *src = in + 16 + (5 - limb) * 8;
        for (int b = 0; b < 8; b++) w = (w << 8) | src[b];
        raw.l[limb] = w;
    }
    /* value < p  <=>  raw - p borrows */
    uint64_t borrow = 0, nz = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)raw.l[i] - ORA_P[i] - borrow;
        borrow = (uint64_t)(d >> 64) & 1;
        nz |= raw.l[i];
    }
    if (!borrow) return -1;
    fp_to_mont(out, &raw);
    return nz != 0;
}
static void ora_fp_to_bytes(uint8_t *out, const fp *a) {
    fp raw;
    fp_from_mont(&raw, a);
    memset(out, 0, 16);
    for (int limb = 0; limb < 6; limb++) {
        uint64_t w = raw.l[limb];
        uint8_t *dst = out + 16 + (5 - limb) * 8;
        for (int b = 7; b >= 0; b--) { dst[b] = (uint8_t)w; w >>= 8; }
    }
}
static int ora_decode_g1(g1_aff *out, const uint8_t *in) {
    int sx = ora_fp_from_bytes(&out->x, in);
    int sy = ora_fp_from_bytes(&out->y, in + 64);
    if (sx < 0 || sy < 0) return ORA_INVALID_ELEMENT;
    if (sx == 0 && sy == 0) return ORA_SUCCESS;
    if (!g1_aff_on_curve(out)) return ORA_POINT_NOT_ON_CURVE;
    return ORA_SUCCESS;
}
static void ora_encode_g1(uint8_t *out, const g1_aff *a) {
    ora_fp_to_bytes(out, &a->x);
    ora_fp_to_bytes(out + 64, &a->y);
}
static int ora_fp2_from_bytes(fp2 *out, const uint8_t *in) {
    int s0 = ora_fp_from_bytes(&out->c0, in);
    int s1 = ora_fp_from_bytes(&out->c1, in + 64);
    if (s0 < 0 || s1 < 0) return -1;
    return s0 | s1;
}
static void ora_fp2_to_bytes(uint8_t *out, const fp2 *a) {
    ora_fp_to_bytes(out, &a->c0);
    ora_fp_to_bytes(out + 64, &a->c1);
}
static int ora_decode_g2(g2_aff *out, const uint8_t *in) {
    int sx = ora_fp2_from_bytes(&out->x, in);
    int sy = ora_fp2_from_bytes(&out->y, in + 128);
    if (sx < 0 || sy < 0) return ORA_INVALID_ELEMENT;
    if (sx == 0 && sy == 0) return ORA_SUCCESS;
    if (!g2_aff_on_curve(out)) return ORA_POINT_NOT_ON_CURVE;
    return ORA_SUCCESS;
}
static void ora_encode_g2(uint8_t *out, const g2_aff *a) {
    ora_fp2_to_bytes(out, &a->x);
    ora_fp2_to_bytes(out + 128, &a->y);
}
/* 32 big-endian bytes -> 32 little-endian bytes; never fails, no reduction (:417-420) */
static void ora_decode_scalar(uint8_t out_le[32], const uint8_t *in) {
    for (int i = 0; i < 32; i++) out_le[i] = in[31 - i];
}

/* ------------------------------------------------------------------ Bos-Coster heap */
typedef struct { uint64_t k[4]; uint32_t base; } ora_heap_item;

static inline int sc_less(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] < b[i];
    }
    return 0;
}
static inline void sc_sub(uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
static inline int sc_bits(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i]) return 64 * (i + 1) - __builtin_clzll(a[i]);
    }
    return 0;
}
static inline void sc_to_le_bytes(uint8_t out[32], const uint64_t a[4]) {
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(a[i / 8] >> (8 * (i % 8)));
}
/* move the hole at `pos` up towards `stop` while the parent is smaller than `item` */
static void heap_bubble_up(ora_heap_item *h, int stop, int pos, const ora_heap_item *item) {
    while (pos > stop) {
        int parent = (pos - 1) >> 1;
        if (!sc_less(h[parent].k, item->k)) break;
        h[pos] = h[parent];
        pos = parent;
    }
    h[pos] = *item;
}
/* restore the max-heap below `start`: walk the larger child down to a leaf, then bubble the
 * displaced item back up (the bottom-up strategy of src/eip2537.c:126-144) */
static void heap_fix_down(ora_heap_item *h, int size, int start) {
    ora_heap_item item = h[start];
    int pos = start, child = 2 * start + 1;
    while (child < size) {
        int right = child + 1;
        if (right < size && !sc_less(h[right].k, h[child].k)) child = right;
        h[pos] = h[child];
        pos = child;
        child = 2 * pos + 1;
    }
    heap_bubble_up(h, start, pos, &item);
}
static void heap_build(ora_heap_item *h, int size) {
    for (int i = (size - 1) / 2; i >= 0; i--) heap_fix_down(h, size, i);
}

#define DEFINE_MSM(G, REC, PTB, OUTB, DECODE, ENCODE)                                              \
/* one Bos-Coster step; returns 0 when the second-largest scalar is zero (:154-205) */             \
static int G##_bc_step(G##_jac *side, G##_jac *bases, ora_heap_item *h, int size) {               \
    ora_heap_item *second = &h[1];                                                                 \
    if (size > 2 && sc_less(h[1].k, h[2].k)) second = &h[2];                                       \
    int second_bits = sc_bits(second->k);                                                          \
    if (second_bits == 0) return 0;                                                                \
    int top_bits = sc_bits(h[0].k);                                                                \
    if (top_bits - second_bits > 6) {                                                              \
        uint8_t kb[32];                                                                            \
        sc_to_le_bytes(kb, h[0].k);                                                                \
        G##_jac *b = &bases[h[0].base];                                                            \
        if (!G##_is_inf(side)) {                                                                   \
            G##_mult(b, b, kb, top_bits);                                                          \
            G##_add(side, side, b);                                                                \
        } else {                                                                                   \
            G##_mult(side, b, kb, top_bits);                                                       \
        }                                                                                          \
        memset(h[0].k, 0, sizeof h[0].k);                                                          \
    } else {                                                                                       \
        sc_sub(h[0].k, second->k);                                                                 \
        G##_add(&bases[second->base], &bases[second->base], &bases[h[0].base]);                    \
    }                                                                                              \
    heap_fix_down(h, size, 0);                                                                     \
    return 1;                                                                                      \
}                                                                                                  \
EXPORT int oracle_bls12_##G##mul(uint8_t *out, const uint8_t *in, size_t in_len) {                 \
    if (in_len != REC) return ORA_INVALID_LENGTH;                                                  \
    G##_aff a;                                                                                     \
    int rc = DECODE(&a, in);                                                                       \
    if (rc != ORA_SUCCESS) return rc;                                                              \
    uint8_t k[32];                                                                                 \
    ora_decode_scalar(k, in + PTB);                                                                \
    G##_jac p;                                                                                     \
    G##_from_affine(&p, &a);                                                                       \
    G##_mult(&p, &p, k, 256);                                                                      \
    G##_to_affine(&a, &p);                                                                         \
    ENCODE(out, &a);                                                                               \
    return ORA_SUCCESS;                                                                            \
}                                                                                                  \
EXPORT int oracle_bls12_##G##multiexp_naive(uint8_t *out, const uint8_t *in, size_t in_len) {      \
    if (in_len == 0 || in_len % REC) return ORA_INVALID_LENGTH;                                    \
    size_t n = in_len / REC;                                                                       \
    G##_jac acc;                                                                                   \
    G##_set_inf(&acc);                                                                             \
    for (size_t i = 0; i < n; i++, in += REC) {                                                    \
        G##_aff a;                                                                                 \
        int rc = DECODE(&a, in);                                                                   \
        if (rc != ORA_SUCCESS) return rc;                                                          \
        uint8_t k[32];                                                                             \
        ora_decode_scalar(k, in + PTB);                                                            \
        G##_jac p;                                                                                 \
        G##_from_affine(&p, &a);                                                                   \
        G##_mult(&p, &p, k, 256);                                                                  \
        G##_add(&acc, &acc, &p);                                                                   \
    }                                                                                              \
    G##_aff r;                                                                                     \
    G##_to_affine(&r, &acc);                                                                       \
    ENCODE(out, &r);                                                                               \
    return ORA_SUCCESS;                                                                            \
}                                                                                                  \
EXPORT int oracle_bls12_##G##multiexp_bc(uint8_t *out, const uint8_t *in, size_t in_len) {         \
    if (in_len == 0 || in_len % REC) return ORA_INVALID_LENGTH;                                    \
    size_t n = in_len / REC;                                                                       \
    G##_jac *bases = malloc(n * sizeof *bases);                                                    \
    if (!bases) return ORA_MEMORY_ERROR;                                                           \
    ora_heap_item *heap = malloc(n * sizeof *heap);                                                \
    if (!heap) { free(bases); return ORA_MEMORY_ERROR; }                                           \
    for (size_t i = 0; i < n; i++, in += REC) {                                                    \
        G##_aff a;                                                                                 \
        int rc = DECODE(&a, in);                                                                   \
        if (rc != ORA_SUCCESS) { free(bases); free(heap); return rc; }                             \
        G##_from_affine(&bases[i], &a);                                                            \
        uint8_t k[32];                                                                             \
        ora_decode_scalar(k, in + PTB);                                                            \
        memcpy(heap[i].k, k, 32);                                                                  \
        heap[i].base = (uint32_t)i;                                                                \
    }                                                                                              \
    heap_build(heap, (int)n);                                                                      \
    G##_jac side, res;                                                                             \
    G##_set_inf(&side);                                                                            \
    if (n > 1) { while (G##_bc_step(&side, bases, heap, (int)n)) { } }                             \
    uint8_t kb[32];                                                                                \
    sc_to_le_bytes(kb, heap[0].k);                                                                 \
    G##_mult(&res, &bases[heap[0].base], kb, sc_bits(heap[0].k));                                  \
    if (!G##_is_inf(&side)) G##_add(&res, &res, &side);                                            \
    G##_aff r;                                                                                     \
    G##_to_affine(&r, &res);                                                                       \
    ENCODE(out, &r);                                                                               \
    free(bases);                                                                                   \
    free(heap);                                                                                    \
    return ORA_SUCCESS;                                                                            \
}                                                                                                  \
EXPORT int oracle_bls12_##G##multiexp(uint8_t *out, const uint8_t *in, size_t in_len) {            \
    if (in_len == 0 || in_len % REC) return ORA_INVALID_LENGTH;                                    \
    size_t n = in_len / REC;                                                                       \
    if (n == 1) return oracle_bls12_##G##mul(out, in, in_len);                                     \
    if (n <= 4) return oracle_bls12_##G##multiexp_naive(out, in, in_len);                          \
    return oracle_bls12_##G##multiexp_bc(out, in, in_len);                                         \
}                                                                                                  \
EXPORT int oracle_bls12_##G##add(uint8_t *out, const uint8_t *in, size_t in_len) {                 \
    if (in_len != 2 * PTB) return ORA_INVALID_LENGTH;                                              \
    G##_aff a, b;                                                                                  \
    int rc = DECODE(&a, in);                                                                       \
    if (rc != ORA_SUCCESS) return rc;                                                              \
    rc = DECODE(&b, in + PTB);                                                                     \
    if (rc != ORA_SUCCESS) return rc;                                                              \
    G##_jac p;                                                                                     \
    G##_from_affine(&p, &a);                                                                       \
    G##_add_affine(&p, &p, &b);                                                                    \
    G##_to_affine(&a, &p);                                                                         \
    ENCODE(out, &a);                                                                               \
    return ORA_SUCCESS;                                                                            \
}

DEFINE_MSM(g1, 160, 128, 128, ora_decode_g1, ora_encode_g1)
DEFINE_MSM(g2, 288, 256, 256, ora_decode_g2, ora_encode_g2)

/* ------------------------------------------------------------------ pairing (:1020-1081) */
EXPORT int oracle_bls12_pairing(uint8_t *out, const uint8_t *in, size_t in_len) {
    if (in_len == 0 || in_len % 384) return ORA_INVALID_LENGTH;
    size_t k = in_len / 384;
    fp12 acc, cur;
    fp12_one(&acc);
    for (size_t i = 0; i < k; i++, in += 384) {
        g1_aff p;
        g2_aff q;
        int rc = ora_decode_g1(&p, in);
        if (rc != ORA_SUCCESS) return rc;
        if (!g1_in_subgroup(&p)) return ORA_POINT_NOT_IN_SUBGROUP;
        rc = ora_decode_g2(&q, in + 128);
        if (rc != ORA_SUCCESS) return rc;
        if (!g2_in_subgroup(&q)) return ORA_POINT_NOT_IN_SUBGROUP;
        if (i > 0) {
            ora_miller_loop(&cur, &q, &p);
            fp12_mul(&acc, &acc, &cur);
        } else {
            ora_miller_loop(&acc, &q, &p);
        }
    }
    ora_final_exp(&acc, &acc);
    memset(out, 0, 32);
    if (fp12_is_one(&acc)) out[31] = 1;
    return ORA_SUCCESS;
}


/* ------------------------------------------------------------------ map to curve (:1094-1165) */
static int fp_sgn0(const fp *a) { fp r; fp_from_mont(&r, a); return (int)(r.l[0] & 1); }
static int fp2_sgn0(const fp2 *a) {
    fp r0, r1;
    fp_from_mont(&r0, &a->c0);
    fp_from_mont(&r1, &a->c1);
    int s0 = (int)(r0.l[0] & 1), z0 = fp_is_zero(&r0), s1 = (int)(r1.l[0] & 1);
    return s0 | (z0 & s1);
}
/* square root in Fp2 (p = 3 mod 4), "complex" method; returns 1 iff a is a square */
static int fp2_sqrt(fp2 *r, const fp2 *a) {
    fp n, t, x0, x1, two_inv, nn;
    if (fp_is_zero(&a->c1)) {
        if (fp_sqrt(&x0, &a->c0)) { r->c0 = x0; fp_zero(&r->c1); return 1; }
        fp_neg(&t, &a->c0);
        fp_sqrt(&x0, &t);                           /* -a0 is a square when a0 is not */
        fp_zero(&r->c0); r->c1 = x0;
        return 1;
    }
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    if (!fp_sqrt(&n, &n)) return 0;
    fp_one(&two_inv); fp_dbl(&two_inv, &two_inv); fp_inv(&two_inv, &two_inv);
    for (int k = 0; k < 2; k++) {
        if (k == 0) nn = n; else fp_neg(&nn, &n);
        fp_add(&t, &a->c0, &nn);
        fp_mul(&t, &t, &two_inv);
        if (!fp_sqrt(&x0, &t) || fp_is_zero(&x0)) continue;
        fp_dbl(&x1, &x0);
        fp_inv(&x1, &x1);
        fp_mul(&x1, &x1, &a->c1);
        fp2 cand = {x0, x1}, chk;
        fp2_sqr(&chk, &cand);
        if (fp2_eq(&chk, a)) { *r = cand; return 1; }
    }
    return 0;
}

#define F fp
#define FN(x) fp_##x
#define PT g1
#define SW(name) ((const fp *)ORA_ISO_G1_##name)
#define SGN0 fp_sgn0
#define SQRT fp_sqrt
#define XNUM_N 12
#define XDEN_N 11
#define YNUM_N 16
#define YDEN_N 16
#include "ora_sswu_tmpl.h"
#undef F
#undef FN
#undef PT
#undef SW
#undef SGN0
#undef SQRT
#undef XNUM_N
#undef XDEN_N
#undef YNUM_N
#undef YDEN_N

#define F fp2
#define FN(x) fp2_##x
#define PT g2
#define SW(name) ((const fp2 *)ORA_ISO_G2_##name)
#define SGN0 fp2_sgn0
#define SQRT fp2_sqrt
#define XNUM_N 4
#define XDEN_N 3
#define YNUM_N 4
#define YDEN_N 4
#include "ora_sswu_tmpl.h"
#undef F
#undef FN
#undef PT
#undef SW
#undef SGN0
#undef SQRT
#undef XNUM_N
#undef XDEN_N
#undef YNUM_N
#undef YDEN_N

EXPORT int oracle_bls12_map_fp_to_g1(uint8_t *out, const uint8_t *in, size_t in_len) {
    if (in_len != 64) return ORA_INVALID_LENGTH;
    fp u;
    if (ora_fp_from_bytes(&u, in) < 0) return ORA_INVALID_ELEMENT;
    g1_aff a;
    g1_map_to_curve(&a, &u);
    g1_jac p;
    g1_from_affine(&p, &a);
    g1_mult(&p, &p, (const uint8_t *)ORA_ISO_H_EFF_G1, 64);          /* clear cofactor: h_eff = 1 - z */
    g1_to_affine(&a, &p);
    ora_encode_g1(out, &a);
    return ORA_SUCCESS;
}
EXPORT int oracle_bls12_map_fp2_to_g2(uint8_t *out, const uint8_t *in, size_t in_len) {
    if (in_len != 128) return ORA_INVALID_LENGTH;
    fp2 u;
    if (ora_fp2_from_bytes(&u, in) < 0) return ORA_INVALID_ELEMENT;
    g2_aff a;
    g2_map_to_curve(&a, &u);
    g2_jac p;
    g2_from_affine(&p, &a);
    g2_mult(&p, &p, (const uint8_t *)ORA_ISO_H_EFF_G2, ORA_ISO_H_EFF_G2_BITS);   /* h_eff = h2 (3 z^2 - 3) */
    g2_to_affine(&a, &p);
    ora_encode_g2(out, &a);
    return ORA_SUCCESS;
}

/* ------------------------------------------------------------------ gas (:1168-1271) */
static const uint64_t ORA_DISCOUNT[128] = {
    1200, 888, 764, 641, 594, 547, 500, 453, 438, 423, 408, 394, 379, 364, 349, 334,
    330, 326, 322, 318, 314, 310, 306, 302, 298, 294, 289, 285, 281, 277, 273, 269,
    268, 266, 265, 263, 262, 260, 259, 257, 256, 254, 253, 251, 250, 248, 247, 245,
    244, 242, 241, 239, 238, 236, 235, 233, 232, 231, 229, 228, 226, 225, 223, 222,
    221, 220, 219, 219, 218, 217, 216, 216, 215, 214, 213, 213, 212, 211, 211, 210,
    209, 208, 208, 207, 206, 205, 205, 204, 203, 202, 202, 201, 200, 199, 199, 198,
    197, 196, 196, 195, 194, 193, 193, 192, 191, 191, 190, 189, 188, 188, 187, 186,
    185, 185, 184, 183, 182, 182, 181, 180, 179, 179, 178, 177, 176, 176, 175, 174};
static uint64_t ora_msm_gas(uint64_t len, uint64_t rec, uint64_t mul_gas) {
    uint64_t k = len / rec;
    if (k == 0) return 0;
    uint64_t d = ORA_DISCOUNT[k < 128 ? k - 1 : 127];
    return k * mul_gas * d / 1000;
}
EXPORT uint64_t oracle_g1multiexp_gas(uint64_t len) { return ora_msm_gas(len, 160, 12000); }
EXPORT uint64_t oracle_g2multiexp_gas(uint64_t len) { return ora_msm_gas(len, 288, 55000); }
EXPORT uint64_t oracle_pairing_gas(uint64_t len) {
    uint64_t k = len / 384;
    return k ? 115000 + 23000 * k : 0;
}

/* ------------------------------------------------------------------ context baseline: all-cores Pippenger
 * NOT the reference's algorithm (the reference is single-threaded Bos-Coster, src/eip2537.c:619-708,
 * README.md:15-17): a plain bucket method over unsigned c-bit windows, windows dealt over `threads`
 * pthreads, so that bench.py can print "what the host's cores could do" next to the 1-thread port
 * (SURVEY.md 8d).  Same input, same canonical output as oracle_bls12_g1multiexp. */
#include <pthread.h>
typedef struct {
    const g1_aff *pts; const uint8_t *scal; size_t n, stride; int c, W, first, step; g1_jac *winsum;
} ora_pip_job;
static void *ora_pip_worker(void *arg) {
    ora_pip_job *j = arg;
    const size_t nb = ((size_t)1 << j->c) - 1;
    g1_jac *bk = malloc(nb * sizeof *bk);
    if (!bk) return (void *)1;
    for (int w = j->first; w < j->W; w += j->step) {
        for (size_t b = 0; b < nb; b++) g1_set_inf(&bk[b]);
        const int lo = w * j->c;
        for (size_t i = 0; i < j->n; i++) {
            const uint8_t *k = j->scal + i * j->stride;          /* 32 bytes little-endian */
            uint32_t d = 0;
            for (int b = 0; b < j->c && lo + b < 256; b++) d |= (uint32_t)((k[(lo + b) >> 3] >> ((lo + b) & 7)) & 1) << b;
            if (d && !g1_aff_is_inf(&j->pts[i])) g1_add_affine(&bk[d - 1], &bk[d - 1], &j->pts[i]);
        }
        g1_jac run, sum;
        g1_set_inf(&run);
        g1_set_inf(&sum);
        for (size_t b = nb; b-- > 0;) { g1_add(&run, &run, &bk[b]); g1_add(&sum, &sum, &run); }
        j->winsum[w] = sum;
    }
    free(bk);
    return NULL;
}
EXPORT int oracle_g1_pippenger_mt(uint8_t *out, const uint8_t *in, size_t in_len, int threads) {
    if (in_len == 0 || in_len % 160) return ORA_INVALID_LENGTH;
    const size_t n = in_len / 160;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    int c = 4;
    while (c < 16 && ((size_t)1 << (c + 4)) < n) c++;             /* ~n/16 buckets per window */
    const int W = (256 + c - 1) / c;
    g1_aff *pts = malloc(n * sizeof *pts);
    uint8_t *scal = malloc(n * 32);
    g1_jac *winsum = malloc((size_t)W * sizeof *winsum);
    int rc = (pts && scal && winsum) ? ORA_SUCCESS : ORA_MEMORY_ERROR;
    for (size_t i = 0; i < n && rc == ORA_SUCCESS; i++) {
        rc = ora_decode_g1(&pts[i], in + i * 160);
        ora_decode_scalar(scal + i * 32, in + i * 160 + 128);
    }
    if (rc == ORA_SUCCESS) {
        pthread_t th[256];
        ora_pip_job jobs[256];
        for (int t = 0; t < threads; t++) {
            jobs[t] = (ora_pip_job){pts, scal, n, 32, c, W, t, threads, winsum};
            if (t > 0 && pthread_create(&th[t], NULL, ora_pip_worker, &jobs[t]) != 0) rc = ORA_MEMORY_ERROR;
        }
        if (ora_pip_worker(&jobs[0]) != NULL) rc = ORA_MEMORY_ERROR;
        for (int t = 1; t < threads; t++) { void *r = NULL; pthread_join(th[t], &r); if (r) rc = ORA_MEMORY_ERROR; }
    }
    if (rc == ORA_SUCCESS) {
        g1_jac acc;
        g1_set_inf(&acc);
        for (int w = W - 1; w >= 0; w--) {
            for (int d = 0; d < c; d++) g1_dbl(&acc, &acc);
            g1_add(&acc, &acc, &winsum[w]);
        }
        g1_aff r;
        g1_to_affine(&r, &acc);
        ora_encode_g1(out, &r);
    }
    free(pts); free(scal); free(winsum);
    return rc;
}

/* ------------------------------------------------------------------ test helpers */
/* slow-definition subgroup tests, to validate the endomorphism tests */
EXPORT int oracle_g1_in_subgroup(const uint8_t in[128], int slow) {
    g1_aff a;
    int rc = ora_decode_g1(&a, in);
    if (rc) return -rc;
    return slow ? g1_in_subgroup_slow(&a) : g1_in_subgroup(&a);
}
EXPORT int oracle_g2_in_subgroup(const uint8_t in[256], int slow) {
    g2_aff a;
    int rc = ora_decode_g2(&a, in);
    if (rc) return -rc;
    return slow ? g2_in_subgroup_slow(&a) : g2_in_subgroup(&a);
}

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/*
 * Synthetic MSM input: P_i = [a + i*b]G for the group generator G, scalars k_i =
 * four big-endian SplitMix64 outputs (uniform 256-bit, not reduced).  Points are produced by
 * repeated addition and converted with one batched inversion.  a_le / b_le are 32-byte LE.
 * Writes n records of (point || scalar) in EIP encoding.
 */
#define DEFINE_GEN(G, F, FNP, REC, PTB, ENCODE, GX, GY, SETXY)                                     \
EXPORT int oracle_gen_##G##_msm_input(uint8_t *out, size_t n, const uint8_t a_le[32],              \
                                      const uint8_t b_le[32], uint64_t seed) {                     \
    G##_aff gen;                                                                                   \
    SETXY;                                                                                         \
    G##_jac g, cur, step;                                                                          \
    G##_from_affine(&g, &gen);                                                                     \
    G##_mult(&cur, &g, a_le, 256);                                                                 \
    G##_mult(&step, &g, b_le, 256);                                                                \
    G##_jac *pts = malloc(n * sizeof *pts);                                                        \
    F *pref = malloc(n * sizeof *pref);                                                            \
    if (!pts || !pref) { free(pts); free(pref); return ORA_MEMORY_ERROR; }                         \
    F run;                                                                                         \
    FNP##one(&run);                                                                                \
    for (size_t i = 0; i < n; i++) {                                                               \
        pts[i] = cur;                                                                              \
        pref[i] = run;                                                                             \
        if (!G##_is_inf(&cur)) FNP##mul(&run, &run, &cur.z);                                       \
        G##_add(&cur, &cur, &step);                                                                \
    }                                                                                              \
    F inv;                                                                                         \
    FNP##inv(&inv, &run);                                                                          \
    uint64_t s = seed;                                                                             \
    for (size_t i = n; i-- > 0;) {                                                                 \
        G##_aff a;                                                                                 \
        if (G##_is_inf(&pts[i])) {                                                                 \
            memset(&a, 0, sizeof a);                                                               \
        } else {                                                                                   \
            F zi, zi2, zi3;                                                                        \
            FNP##mul(&zi, &inv, &pref[i]);                                                         \
            FNP##mul(&inv, &inv, &pts[i].z);                                                       \
            FNP##sqr(&zi2, &zi);                                                                   \
            FNP##mul(&zi3, &zi2, &zi);                                                             \
            FNP##mul(&a.x, &pts[i].x, &zi2);                                                       \
            FNP##mul(&a.y, &pts[i].y, &zi3);                                                       \
        }                                                                                          \
        ENCODE(out + i * REC, &a);                                                                 \
    }                                                                                              \
    for (size_t i = 0; i < n; i++) {                                                               \
        uint8_t *k = out + i * REC + PTB;                                                          \
        for (int w = 0; w < 4; w++) {                                                              \
            uint64_t v = splitmix64(&s);                                                           \
            for (int b = 0; b < 8; b++) k[8 * w + b] = (uint8_t)(v >> (56 - 8 * b));               \
        }                                                                                          \
    }                                                                                              \
    free(pts);                                                                                     \
    free(pref);                                                                                    \
    return ORA_SUCCESS;                                                                            \
}

DEFINE_GEN(g1, fp, fp_, 160, 128, ora_encode_g1, ORA_G1_X, ORA_G1_Y,
           (fp_set(&gen.x, ORA_G1_X), fp_set(&gen.y, ORA_G1_Y)))
DEFINE_GEN(g2, fp2, fp2_, 288, 256, ora_encode_g2, ORA_G2_X, ORA_G2_Y,
           (fp2_set(&gen.x, ORA_G2_X), fp2_set(&gen.y, ORA_G2_Y)))

/*
 * Synthetic pairing input: pairs ([a0 + i*a1]G1, [b0 + i*b1]G2), i < k.
 * The caller (Python, big integers) fixes up the last pair so that the product is 1.
 */
EXPORT int oracle_gen_pairing_input(uint8_t *out, size_t k, const uint8_t a0[32], const uint8_t a1[32],
                                    const uint8_t b0[32], const uint8_t b1[32]) {
    uint8_t *t1 = malloc(k * 160), *t2 = malloc(k * 288);
    if (!t1 || !t2) { free(t1); free(t2); return ORA_MEMORY_ERROR; }
    oracle_gen_g1_msm_input(t1, k, a0, a1, 1);
    oracle_gen_g2_msm_input(t2, k, b0, b1, 1);
    for (size_t i = 0; i < k; i++) {
        memcpy(out + i * 384, t1 + i * 160, 128);
        memcpy(out + i * 384 + 128, t2 + i * 288, 256);
    }
    free(t1);
    free(t2);
    return ORA_SUCCESS;
}

/* raw Miller loop / final exponentiation on encoded points, for cross-checks in tests:
 * out = 576 bytes = 12 Fp coefficients (c0.a0.c0, c0.a0.c1, c0.a1.c0, ...) 48-byte big-endian */
EXPORT int oracle_pairing_fp12(uint8_t out[576], const uint8_t in[384], int do_final_exp) {
    g1_aff p;
    g2_aff q;
    int rc = ora_decode_g1(&p, in);
    if (rc) return rc;
    rc = ora_decode_g2(&q, in + 128);
    if (rc) return rc;
    fp12 f;
    ora_miller_loop(&f, &q, &p);
    if (do_final_exp) ora_final_exp(&f, &f);
    for (int k = 0; k < 6; k++) {
        uint8_t tmp[128];
        ora_fp2_to_bytes(tmp, fp12_coeff(&f, k));
        memcpy(out + k * 96, tmp + 16, 48);
        memcpy(out + k * 96 + 48, tmp + 80, 48);
    }
    return ORA_SUCCESS;
}
