/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not linked into the product library.
 * PARITY UNPINNED (see oracle/README.md): blst is not vendored with the reference and the
 * reference's known-answer files are absent; this is a from-scratch CPU restatement.
 *
 * Field tower for BLS12-381 on 6 x 64-bit Montgomery limbs (R = 2^384), the same
 * representation the reference obtains from blst (blst_fp = 6 x limb_t, src/eip2537.c:276-301).
 *   Fp2  = Fp[u]/(u^2+1)        Fp6 = Fp2[v]/(v^3-(1+u))       Fp12 = Fp6[w]/(w^2-v)
 * Stands in for: blst_fp_add / blst_fp_to / blst_fp_from / blst_bendian_from_fp (eip2537.c:287,301,316)
 * and the Fp12 arithmetic behind blst_fp12_mul / blst_final_exp (eip2537.c:1061,1070).
 */
#pragma once
#include <stdint.h>
#include <string.h>
#include "ora_constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 a0, a1, a2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

/* ---------------------------------------------------------------- Fp */
static inline void fp_set(fp *r, const uint64_t w[6]) { memcpy(r->l, w, 48); }
static inline void fp_zero(fp *r) { memset(r, 0, sizeof *r); }
static inline void fp_one(fp *r) { fp_set(r, ORA_ONE); }
static inline int fp_is_zero(const fp *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0;
}
static inline int fp_eq(const fp *a, const fp *b) {
    uint64_t d = 0;
    for (int i = 0; i < 6; i++) d |= a->l[i] ^ b->l[i];
    return d == 0;
}
/* r = a - p if a >= p else a   (a < 2p, carry = bit 384 of a) */
static inline void fp_cond_sub_p(fp *r, const uint64_t a[6], uint64_t carry) {
    uint64_t t[6], borrow = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a[i] - ORA_P[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    /* keep the difference iff no net borrow out of (carry:a) - p */
    uint64_t use_t = (carry | (borrow ^ 1)) ? ~0ULL : 0ULL;
    for (int i = 0; i < 6; i++) r->l[i] = (t[i] & use_t) | (a[i] & ~use_t);
}
static inline void fp_add(fp *r, const fp *a, const fp *b) {
    uint64_t t[6], c = 0;
    for (int i = 0; i < 6; i++) {
        u128 s = (u128)a->l[i] + b->l[i] + c;
        t[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
    }
    fp_cond_sub_p(r, t, c);
}
static inline void fp_sub(fp *r, const fp *a, const fp *b) {
    uint64_t t[6], borrow = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    uint64_t mask = borrow ? ~0ULL : 0ULL, c = 0;
    for (int i = 0; i < 6; i++) {
        u128 s = (u128)t[i] + (ORA_P[i] & mask) + c;
        r->l[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
    }
}
static inline void fp_neg(fp *r, const fp *a) {
    fp z;
    fp_zero(&z);
    fp_sub(r, &z, a);
}
static inline void fp_dbl(fp *r, const fp *a) { fp_add(r, a, a); }

/* Montgomery product a*b/R mod p, coarsely-integrated operand scanning. */
static inline void fp_mul(fp *r, const fp *a, const fp *b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        uint64_t c = 0, bi = b->l[i];
        for (int j = 0; j < 6; j++) {
            u128 s = (u128)a->l[j] * bi + t[j] + c;
            t[j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        u128 s = (u128)t[6] + c;
        t[6] = (uint64_t)s;
        t[7] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * ORA_N0;
        s = (u128)m * ORA_P[0] + t[0];
        c = (uint64_t)(s >> 64);
        for (int j = 1; j < 6; j++) {
            s = (u128)m * ORA_P[j] + t[j] + c;
            t[j - 1] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        s = (u128)t[6] + c;
        t[5] = (uint64_t)s;
        t[6] = t[7] + (uint64_t)(s >> 64);
    }
    fp_cond_sub_p(r, t, t[6]);
}
static inline void fp_sqr(fp *r, const fp *a) { fp_mul(r, a, a); }
/* r = a^e, e little-endian 64-bit words (plain integer), MSB-first square and multiply */
static inline void fp_pow(fp *r, const fp *a, const uint64_t *e, int nwords) {
    fp acc, base = *a;
    fp_one(&acc);
    for (int i = nwords * 64 - 1; i >= 0; i--) {
        fp_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
    }
    *r = acc;
}
/* inverse by Fermat; inv(0) = 0, which makes to_affine(infinity) = (0,0) like blst_p1_to_affine */
static inline void fp_inv(fp *r, const fp *a) { fp_pow(r, a, ORA_P_MINUS_2, 6); }
static inline void fp_to_mont(fp *r, const fp *a) {
    fp rr;
    fp_set(&rr, ORA_RR);
    fp_mul(r, a, &rr);
}
static inline void fp_from_mont(fp *r, const fp *a) {
    fp one = {{1, 0, 0, 0, 0, 0}};
    fp_mul(r, a, &one);
}
/* sqrt for p = 3 mod 4; returns 1 if a is a square */
static inline int fp_sqrt(fp *r, const fp *a) {
    fp s, t;
    fp_pow(&s, a, ORA_P_PLUS_1_DIV_4, 6);
    fp_sqr(&t, &s);
    int ok = fp_eq(&t, a);          /* before writing r: r may alias a */
    *r = s;
    return ok;
}

/* ---------------------------------------------------------------- Fp2 */
static inline void fp2_zero(fp2 *r) { memset(r, 0, sizeof *r); }
static inline void fp2_one(fp2 *r) { fp_one(&r->c0); fp_zero(&r->c1); }
static inline int fp2_is_zero(const fp2 *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static inline int fp2_eq(const fp2 *a, const fp2 *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static inline void fp2_add(fp2 *r, const fp2 *a, const fp2 *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static inline void fp2_sub(fp2 *r, const fp2 *a, const fp2 *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static inline void fp2_neg(fp2 *r, const fp2 *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static inline void fp2_dbl(fp2 *r, const fp2 *a) { fp2_add(r, a, a); }
static inline void fp2_conj(fp2 *r, const fp2 *a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static inline void fp2_mul(fp2 *r, const fp2 *a, const fp2 *b) {
    fp t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_add(&t2, &a->c0, &a->c1);
    fp_add(&t3, &b->c0, &b->c1);
    fp_mul(&t2, &t2, &t3);
    fp_sub(&t2, &t2, &t0);
    fp_sub(&r->c1, &t2, &t1);
    fp_sub(&r->c0, &t0, &t1);
}
static inline void fp2_sqr(fp2 *r, const fp2 *a) {
    fp s, d, m;
    fp_add(&s, &a->c0, &a->c1);
    fp_sub(&d, &a->c0, &a->c1);
    fp_mul(&m, &a->c0, &a->c1);
    fp_mul(&r->c0, &s, &d);
    fp_dbl(&r->c1, &m);
}
static inline void fp2_mul_fp(fp2 *r, const fp2 *a, const fp *s) { fp_mul(&r->c0, &a->c0, s); fp_mul(&r->c1, &a->c1, s); }
/* multiply by xi = 1 + u */
static inline void fp2_mul_xi(fp2 *r, const fp2 *a) {
    fp t;
    fp_sub(&t, &a->c0, &a->c1);
    fp_add(&r->c1, &a->c0, &a->c1);
    r->c0 = t;
}
static inline void fp2_inv(fp2 *r, const fp2 *a) {
    fp n, t;
    fp_sqr(&n, &a->c0);
    fp_sqr(&t, &a->c1);
    fp_add(&n, &n, &t);
    fp_inv(&n, &n);
    fp_mul(&r->c0, &a->c0, &n);
    fp_mul(&t, &a->c1, &n);
    fp_neg(&r->c1, &t);
}
static inline void fp2_set(fp2 *r, const uint64_t w[2][6]) { fp_set(&r->c0, w[0]); fp_set(&r->c1, w[1]); }

/* ---------------------------------------------------------------- Fp6 */
static inline void fp6_zero(fp6 *r) { memset(r, 0, sizeof *r); }
static inline void fp6_one(fp6 *r) { fp6_zero(r); fp_one(&r->a0.c0); }
static inline void fp6_add(fp6 *r, const fp6 *a, const fp6 *b) { fp2_add(&r->a0, &a->a0, &b->a0); fp2_add(&r->a1, &a->a1, &b->a1); fp2_add(&r->a2, &a->a2, &b->a2); }
static inline void fp6_sub(fp6 *r, const fp6 *a, const fp6 *b) { fp2_sub(&r->a0, &a->a0, &b->a0); fp2_sub(&r->a1, &a->a1, &b->a1); fp2_sub(&r->a2, &a->a2, &b->a2); }
static inline void fp6_neg(fp6 *r, const fp6 *a) { fp2_neg(&r->a0, &a->a0); fp2_neg(&r->a1, &a->a1); fp2_neg(&r->a2, &a->a2); }
static inline int fp6_eq(const fp6 *a, const fp6 *b) { return fp2_eq(&a->a0, &b->a0) && fp2_eq(&a->a1, &b->a1) && fp2_eq(&a->a2, &b->a2); }
/* Karatsuba-style 6-multiplication product modulo v^3 = xi */
static inline void fp6_mul(fp6 *r, const fp6 *a, const fp6 *b) {
    fp2 v0, v1, v2, s, t, u, c0, c1, c2;
    fp2_mul(&v0, &a->a0, &b->a0);
    fp2_mul(&v1, &a->a1, &b->a1);
    fp2_mul(&v2, &a->a2, &b->a2);
    /* c0 = v0 + xi((a1+a2)(b1+b2) - v1 - v2) */
    fp2_add(&s, &a->a1, &a->a2);
    fp2_add(&t, &b->a1, &b->a2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v1);
    fp2_sub(&u, &u, &v2);
    fp2_mul_xi(&u, &u);
    fp2_add(&c0, &u, &v0);
    /* c1 = (a0+a1)(b0+b1) - v0 - v1 + xi v2 */
    fp2_add(&s, &a->a0, &a->a1);
    fp2_add(&t, &b->a0, &b->a1);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v1);
    fp2_mul_xi(&s, &v2);
    fp2_add(&c1, &u, &s);
    /* c2 = (a0+a2)(b0+b2) - v0 - v2 + v1 */
    fp2_add(&s, &a->a0, &a->a2);
    fp2_add(&t, &b->a0, &b->a2);
    fp2_mul(&u, &s, &t);
    fp2_sub(&u, &u, &v0);
    fp2_sub(&u, &u, &v2);
    fp2_add(&c2, &u, &v1);
    r->a0 = c0; r->a1 = c1; r->a2 = c2;
}
static inline void fp6_mul_by_v(fp6 *r, const fp6 *a) {
    fp2 t;
    fp2_mul_xi(&t, &a->a2);
    r->a2 = a->a1;
    r->a1 = a->a0;
    r->a0 = t;
}
/* a * (b0 + b1 v) */
static inline void fp6_mul_by_01(fp6 *r, const fp6 *a, const fp2 *b0, const fp2 *b1) {
    fp2 t0, t1, c0, c1, c2;
    fp2_mul(&t0, &a->a2, b1);
    fp2_mul_xi(&t0, &t0);
    fp2_mul(&t1, &a->a0, b0);
    fp2_add(&c0, &t0, &t1);
    fp2_mul(&t0, &a->a0, b1);
    fp2_mul(&t1, &a->a1, b0);
    fp2_add(&c1, &t0, &t1);
    fp2_mul(&t0, &a->a1, b1);
    fp2_mul(&t1, &a->a2, b0);
    fp2_add(&c2, &t0, &t1);
    r->a0 = c0; r->a1 = c1; r->a2 = c2;
}
/* a * (b1 v) */
static inline void fp6_mul_by_1(fp6 *r, const fp6 *a, const fp2 *b1) {
    fp2 c0, c1, c2;
    fp2_mul(&c0, &a->a2, b1);
    fp2_mul_xi(&c0, &c0);
    fp2_mul(&c1, &a->a0, b1);
    fp2_mul(&c2, &a->a1, b1);
    r->a0 = c0; r->a1 = c1; r->a2 = c2;
}
static inline void fp6_inv(fp6 *r, const fp6 *a) {
    fp2 c0, c1, c2, t, u;
    fp2_sqr(&c0, &a->a0);
    fp2_mul(&t, &a->a1, &a->a2);
    fp2_mul_xi(&t, &t);
    fp2_sub(&c0, &c0, &t);
    fp2_sqr(&c1, &a->a2);
    fp2_mul_xi(&c1, &c1);
    fp2_mul(&t, &a->a0, &a->a1);
    fp2_sub(&c1, &c1, &t);
    fp2_sqr(&c2, &a->a1);
    fp2_mul(&t, &a->a0, &a->a2);
    fp2_sub(&c2, &c2, &t);
    fp2_mul(&t, &a->a2, &c1);
    fp2_mul(&u, &a->a1, &c2);
    fp2_add(&t, &t, &u);
    fp2_mul_xi(&t, &t);
    fp2_mul(&u, &a->a0, &c0);
    fp2_add(&t, &t, &u);
    fp2_inv(&t, &t);
    fp2_mul(&r->a0, &c0, &t);
    fp2_mul(&r->a1, &c1, &t);
    fp2_mul(&r->a2, &c2, &t);
}

/* ---------------------------------------------------------------- Fp12 */
static inline void fp12_one(fp12 *r) { fp6_one(&r->c0); fp6_zero(&r->c1); }
static inline int fp12_eq(const fp12 *a, const fp12 *b) { return fp6_eq(&a->c0, &b->c0) && fp6_eq(&a->c1, &b->c1); }
static inline int fp12_is_one(const fp12 *a) { fp12 o; fp12_one(&o); return fp12_eq(a, &o); }
static inline void fp12_mul(fp12 *r, const fp12 *a, const fp12 *b) {
    fp6 t0, t1, s, t, u;
    fp6_mul(&t0, &a->c0, &b->c0);
    fp6_mul(&t1, &a->c1, &b->c1);
    fp6_add(&s, &a->c0, &a->c1);
    fp6_add(&t, &b->c0, &b->c1);
    fp6_mul(&u, &s, &t);
    fp6_sub(&u, &u, &t0);
    fp6_sub(&r->c1, &u, &t1);
    fp6_mul_by_v(&t1, &t1);
    fp6_add(&r->c0, &t0, &t1);
}
static inline void fp12_sqr(fp12 *r, const fp12 *a) {
    /* (c0 + c1 w)^2 = (c0+c1)(c0 + v c1) - t - v t + 2 t w,  t = c0 c1 */
    fp6 t, s, u, vt;
    fp6_mul(&t, &a->c0, &a->c1);
    fp6_add(&s, &a->c0, &a->c1);
    fp6_mul_by_v(&u, &a->c1);
    fp6_add(&u, &u, &a->c0);
    fp6_mul(&s, &s, &u);
    fp6_mul_by_v(&vt, &t);
    fp6_sub(&s, &s, &t);
    fp6_sub(&r->c0, &s, &vt);
    fp6_add(&r->c1, &t, &t);
}
static inline void fp12_conj(fp12 *r, const fp12 *a) { r->c0 = a->c0; fp6_neg(&r->c1, &a->c1); }
static inline void fp12_inv(fp12 *r, const fp12 *a) {
    fp6 t, u;
    fp6_mul(&t, &a->c0, &a->c0);
    fp6_mul(&u, &a->c1, &a->c1);
    fp6_mul_by_v(&u, &u);
    fp6_sub(&t, &t, &u);
    fp6_inv(&t, &t);
    fp6_mul(&r->c0, &a->c0, &t);
    fp6_mul(&u, &a->c1, &t);
    fp6_neg(&r->c1, &u);
}
/* f * ((a0 + a1 v) + (a4 v) w) */
static inline void fp12_mul_by_014(fp12 *r, const fp12 *f, const fp2 *a0, const fp2 *a1, const fp2 *a4) {
    fp6 t0, t1, r0, r1, u;
    fp6_mul_by_01(&t0, &f->c0, a0, a1);
    fp6_mul_by_1(&t1, &f->c1, a4);
    fp6_mul_by_v(&u, &t1);
    fp6_add(&r0, &t0, &u);
    fp6_mul_by_1(&t0, &f->c0, a4);
    fp6_mul_by_01(&t1, &f->c1, a0, a1);
    fp6_add(&r1, &t0, &t1);
    r->c0 = r0; r->c1 = r1;
}
static inline fp2 *fp12_coeff(fp12 *a, int k) {
    fp2 *tab[6] = {&a->c0.a0, &a->c0.a1, &a->c0.a2, &a->c1.a0, &a->c1.a1, &a->c1.a2};
    return tab[k];
}
static inline void fp12_frob(fp12 *r, const fp12 *a) {
    fp12 t = *a;
    for (int k = 0; k < 6; k++) {
        fp2 c, g;
        fp2_conj(&c, fp12_coeff(&t, k));
        fp2_set(&g, ORA_FROB1[k]);
        fp2_mul(fp12_coeff(r, k), &c, &g);
    }
}
static inline void fp12_frob2(fp12 *r, const fp12 *a) {
    fp12 t = *a;
    for (int k = 0; k < 6; k++) {
        fp g;
        fp_set(&g, ORA_FROB2[k]);
        fp2_mul_fp(fp12_coeff(r, k), fp12_coeff(&t, k), &g);
    }
}
