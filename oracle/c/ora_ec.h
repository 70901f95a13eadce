/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  G1 / G2 instantiations, subgroup tests, pairing.
 * Stands in for: blst_p1_* / blst_p2_* (src/eip2537.c:457-518, 745-806), blst_p1_affine_in_g1 /
 * blst_p2_affine_in_g2 (:1041,1051), blst_miller_loop (:1060,1065), blst_final_exp (:1070).
 */
#pragma once
#include "ora_field.h"

static const fp *ora_b1(void) { return (const fp *)ORA_B1; }
static const fp2 *ora_b2(void) { return (const fp2 *)ORA_B2; }

#define F fp
#define FN(x) fp_##x
#define PT g1
#define CURVE_B ora_b1()
#include "ora_ec_tmpl.h"
#undef F
#undef FN
#undef PT
#undef CURVE_B

#define F fp2
#define FN(x) fp2_##x
#define PT g2
#define CURVE_B ora_b2()
#include "ora_ec_tmpl.h"
#undef F
#undef FN
#undef PT
#undef CURVE_B

/* ---- subgroup membership (exact r-torsion test; infinity is a member) ---------------- */
/* slow definition: [r]P == infinity */
static inline int g1_in_subgroup_slow(const g1_aff *a) {
    g1_jac p, t;
    uint8_t k[32];
    for (int i = 0; i < 32; i++) k[i] = (uint8_t)(ORA_R_ORDER[i / 8] >> (8 * (i % 8)));
    g1_from_affine(&p, a);
    g1_mult(&t, &p, k, 255);
    return g1_is_inf(&t);
}
static inline int g2_in_subgroup_slow(const g2_aff *a) {
    g2_jac p, t;
    uint8_t k[32];
    for (int i = 0; i < 32; i++) k[i] = (uint8_t)(ORA_R_ORDER[i / 8] >> (8 * (i % 8)));
    g2_from_affine(&p, a);
    g2_mult(&t, &p, k, 255);
    return g2_is_inf(&t);
}
/* fast: phi(P) == -[|z|]([|z|]P), phi(x,y) = (beta x, y)   (validated against the slow form) */
static inline int g1_in_subgroup(const g1_aff *a) {
    if (g1_aff_is_inf(a)) return 1;
    g1_jac p, t, ph;
    g1_from_affine(&p, a);
    g1_mul_zabs(&t, &p);
    g1_mul_zabs(&t, &t);
    g1_neg(&t, &t);
    ph = p;
    fp_mul(&ph.x, &ph.x, (const fp *)ORA_BETA);
    return g1_eq(&ph, &t);
}
/* fast: psi(Q) == [z]Q = -[|z|]Q,  psi(x,y) = (conj(x) PSI_X, conj(y) PSI_Y) */
static inline int g2_in_subgroup(const g2_aff *a) {
    if (g2_aff_is_inf(a)) return 1;
    g2_jac p, t, ps;
    g2_from_affine(&p, a);
    g2_mul_zabs(&t, &p);
    g2_neg(&t, &t);
    fp2_conj(&ps.x, &a->x);
    fp2_mul(&ps.x, &ps.x, (const fp2 *)ORA_PSI_X);
    fp2_conj(&ps.y, &a->y);
    fp2_mul(&ps.y, &ps.y, (const fp2 *)ORA_PSI_Y);
    fp2_one(&ps.z);
    return g2_eq(&ps, &t);
}

/* ---- optimal-ate Miller loop, f_{|z|,Q}(P) conjugated (z < 0), no final exponentiation --- */
/* Line through the running point T (Jacobian on the twist), scaled by Fp2 factors that the
 * final exponentiation kills:  l = a0 + (a1 xP) v + (a4 yP) v w.                              */
static inline void ora_dbl_step(g2_jac *T, fp2 *a0, fp2 *a1, fp2 *a4) {
    fp2 A, B, C, D, E, Fq, ZZ, t, x3, y3, z3;
    fp2_sqr(&A, &T->x);
    fp2_sqr(&B, &T->y);
    fp2_sqr(&C, &B);
    fp2_add(&t, &T->x, &B);
    fp2_sqr(&t, &t);
    fp2_sub(&t, &t, &A);
    fp2_sub(&t, &t, &C);
    fp2_dbl(&D, &t);
    fp2_dbl(&E, &A);
    fp2_add(&E, &E, &A);
    fp2_sqr(&Fq, &E);
    fp2_sqr(&ZZ, &T->z);
    fp2_dbl(&t, &D);
    fp2_sub(&x3, &Fq, &t);
    fp2_sub(&t, &D, &x3);
    fp2_mul(&y3, &E, &t);
    fp2_dbl(&C, &C); fp2_dbl(&C, &C); fp2_dbl(&C, &C);
    fp2_sub(&y3, &y3, &C);
    fp2_mul(&z3, &T->y, &T->z);
    fp2_dbl(&z3, &z3);
    /* a0 = 3X^3 - 2Y^2,  a1 = -3X^2 Z^2,  a4 = 2YZ^3 */
    fp2_mul(a0, &E, &T->x);
    fp2_dbl(&t, &B);
    fp2_sub(a0, a0, &t);
    fp2_mul(a1, &E, &ZZ);
    fp2_neg(a1, a1);
    fp2_mul(a4, &z3, &ZZ);
    T->x = x3; T->y = y3; T->z = z3;
}
static inline void ora_add_step(g2_jac *T, const g2_aff *Q, fp2 *a0, fp2 *a1, fp2 *a4) {
    fp2 ZZ, U2, S2, H, th, HH, HHH, V, t, x3, y3, z3;
    fp2_sqr(&ZZ, &T->z);
    fp2_mul(&U2, &Q->x, &ZZ);
    fp2_mul(&S2, &ZZ, &T->z);
    fp2_mul(&S2, &S2, &Q->y);
    fp2_sub(&H, &U2, &T->x);
    fp2_sub(&th, &S2, &T->y);
    fp2_sqr(&HH, &H);
    fp2_mul(&HHH, &HH, &H);
    fp2_mul(&V, &T->x, &HH);
    fp2_sqr(&x3, &th);
    fp2_sub(&x3, &x3, &HHH);
    fp2_dbl(&t, &V);
    fp2_sub(&x3, &x3, &t);
    fp2_sub(&t, &V, &x3);
    fp2_mul(&y3, &th, &t);
    fp2_mul(&t, &T->y, &HHH);
    fp2_sub(&y3, &y3, &t);
    fp2_mul(&z3, &T->z, &H);
    /* a0 = th x2 - mu y2,  a1 = -th,  a4 = mu = Z3 */
    fp2_mul(a0, &th, &Q->x);
    fp2_mul(&t, &z3, &Q->y);
    fp2_sub(a0, a0, &t);
    fp2_neg(a1, &th);
    *a4 = z3;
    T->x = x3; T->y = y3; T->z = z3;
}
/* A pair with either point at infinity contributes the identity (EIP-2537 semantics). */
static inline void ora_miller_loop(fp12 *f, const g2_aff *Q, const g1_aff *P) {
    fp12_one(f);
    if (g1_aff_is_inf(P) || g2_aff_is_inf(Q)) return;
    g2_jac T;
    g2_from_affine(&T, Q);
    fp2 a0, a1, a4;
    for (int i = 62; i >= 0; i--) {
        ora_dbl_step(&T, &a0, &a1, &a4);
        fp2_mul_fp(&a1, &a1, &P->x);
        fp2_mul_fp(&a4, &a4, &P->y);
        fp12_sqr(f, f);
        fp12_mul_by_014(f, f, &a0, &a1, &a4);
        if ((ORA_Z_ABS >> i) & 1) {
            ora_add_step(&T, Q, &a0, &a1, &a4);
            fp2_mul_fp(&a1, &a1, &P->x);
            fp2_mul_fp(&a4, &a4, &P->y);
            fp12_mul_by_014(f, f, &a0, &a1, &a4);
        }
    }
    fp12_conj(f, f);
}
/* g^z for g in the cyclotomic subgroup (inverse = conjugate) */
static inline void ora_exp_by_z(fp12 *r, const fp12 *g) {
    fp12 acc = *g, base = *g;
    for (int i = 62; i >= 0; i--) {
        fp12_sqr(&acc, &acc);
        if ((ORA_Z_ABS >> i) & 1) fp12_mul(&acc, &acc, &base);
    }
    fp12_conj(r, &acc);
}
/* f^(3 (p^12-1)/r): easy part, then (z-1)^2 (z+p) (z^2+p^2-1) + 3.  Only "== 1" is observed
 * by the reference (src/eip2537.c:1076) and gcd(3, r) = 1.                                  */
static inline void ora_final_exp(fp12 *r, const fp12 *f) {
    fp12 f1, f2, y0, y1, y2, y3, t;
    fp12_inv(&t, f);
    fp12_conj(&f1, f);
    fp12_mul(&f1, &f1, &t);
    fp12_frob2(&f2, &f1);
    fp12_mul(&f2, &f2, &f1);
    ora_exp_by_z(&y0, &f2);
    fp12_conj(&t, &f2);
    fp12_mul(&y0, &y0, &t);
    ora_exp_by_z(&y1, &y0);
    fp12_conj(&t, &y0);
    fp12_mul(&y1, &y1, &t);
    ora_exp_by_z(&y2, &y1);
    fp12_frob(&t, &y1);
    fp12_mul(&y2, &y2, &t);
    ora_exp_by_z(&y3, &y2);
    ora_exp_by_z(&y3, &y3);
    fp12_frob2(&t, &y2);
    fp12_mul(&y3, &y3, &t);
    fp12_conj(&t, &y2);
    fp12_mul(&y3, &y3, &t);
    fp12_sqr(&t, &f2);
    fp12_mul(&t, &t, &f2);
    fp12_mul(r, &y3, &t);
}
