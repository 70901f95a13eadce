/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Curve arithmetic template, instantiated twice
 * (G1 over Fp, G2 over Fp2) by ora_ec.h.  Jacobian coordinates, Z = 0 is infinity,
 * affine (0,0) is infinity -- the conventions the reference relies on from blst
 * (call sites src/eip2537.c:457-465, 598-610, 661-698).
 *
 * Parameters (macros):  F   field type            FN(x) field function  f##x
 *                       PT  point-name prefix     CURVE_B  pointer to b (Montgomery)
 */
#define EC_CAT_(a, b) a##b
#define EC_CAT(a, b) EC_CAT_(a, b)
#define PF(name) EC_CAT(PT, name)

typedef struct { F x, y, z; } PF(_jac);
typedef struct { F x, y; } PF(_aff);

static inline int PF(_is_inf)(const PF(_jac) *p) { return FN(is_zero)(&p->z); }
static inline void PF(_set_inf)(PF(_jac) *p) { memset(p, 0, sizeof *p); }
static inline int PF(_aff_is_inf)(const PF(_aff) *p) { return FN(is_zero)(&p->x) && FN(is_zero)(&p->y); }

/* blst_pN_from_affine: (x,y) -> (x,y,1); (0,0) -> Z = 0 */
static inline void PF(_from_affine)(PF(_jac) *r, const PF(_aff) *a) {
    r->x = a->x;
    r->y = a->y;
    if (PF(_aff_is_inf)(a)) FN(zero)(&r->z); else FN(one)(&r->z);
}
/* blst_pN_to_affine: one inversion; infinity -> (0,0) because inv(0) = 0 */
static inline void PF(_to_affine)(PF(_aff) *r, const PF(_jac) *p) {
    F zi, zi2, zi3;
    FN(inv)(&zi, &p->z);
    FN(sqr)(&zi2, &zi);
    FN(mul)(&zi3, &zi2, &zi);
    FN(mul)(&r->x, &p->x, &zi2);
    FN(mul)(&r->y, &p->y, &zi3);
}
/* blst_pN_affine_on_curve: y^2 == x^3 + b */
static inline int PF(_aff_on_curve)(const PF(_aff) *a) {
    F l, rr;
    FN(sqr)(&l, &a->y);
    FN(sqr)(&rr, &a->x);
    FN(mul)(&rr, &rr, &a->x);
    FN(add)(&rr, &rr, CURVE_B);
    return FN(eq)(&l, &rr);
}
static inline void PF(_neg)(PF(_jac) *r, const PF(_jac) *p) { r->x = p->x; FN(neg)(&r->y, &p->y); r->z = p->z; }

/* doubling, a = 0 (dbl-2009-l) */
static inline void PF(_dbl)(PF(_jac) *r, const PF(_jac) *p) {
    F A, B, C, D, E, Fq, t, x3, y3, z3;
    FN(sqr)(&A, &p->x);
    FN(sqr)(&B, &p->y);
    FN(sqr)(&C, &B);
    FN(add)(&t, &p->x, &B);
    FN(sqr)(&t, &t);
    FN(sub)(&t, &t, &A);
    FN(sub)(&t, &t, &C);
    FN(dbl)(&D, &t);
    FN(dbl)(&E, &A);
    FN(add)(&E, &E, &A);
    FN(sqr)(&Fq, &E);
    FN(dbl)(&t, &D);
    FN(sub)(&x3, &Fq, &t);
    FN(sub)(&t, &D, &x3);
    FN(mul)(&y3, &E, &t);
    FN(dbl)(&C, &C); FN(dbl)(&C, &C); FN(dbl)(&C, &C);
    FN(sub)(&y3, &y3, &C);
    FN(mul)(&z3, &p->y, &p->z);
    FN(dbl)(&z3, &z3);
    r->x = x3; r->y = y3; r->z = z3;
}
/* complete addition (blst_pN_add_or_double): handles inf, P == Q, P == -Q; r may alias */
static inline void PF(_add)(PF(_jac) *r, const PF(_jac) *p, const PF(_jac) *q) {
    if (PF(_is_inf)(p)) { *r = *q; return; }
    if (PF(_is_inf)(q)) { *r = *p; return; }
    F z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    FN(sqr)(&z1z1, &p->z);
    FN(sqr)(&z2z2, &q->z);
    FN(mul)(&u1, &p->x, &z2z2);
    FN(mul)(&u2, &q->x, &z1z1);
    FN(mul)(&s1, &p->y, &q->z);
    FN(mul)(&s1, &s1, &z2z2);
    FN(mul)(&s2, &q->y, &p->z);
    FN(mul)(&s2, &s2, &z1z1);
    if (FN(eq)(&u1, &u2)) {
        if (FN(eq)(&s1, &s2)) PF(_dbl)(r, p); else PF(_set_inf)(r);
        return;
    }
    FN(sub)(&h, &u2, &u1);
    FN(dbl)(&i, &h);
    FN(sqr)(&i, &i);
    FN(mul)(&j, &h, &i);
    FN(sub)(&rr, &s2, &s1);
    FN(dbl)(&rr, &rr);
    FN(mul)(&v, &u1, &i);
    FN(sqr)(&x3, &rr);
    FN(sub)(&x3, &x3, &j);
    FN(dbl)(&t, &v);
    FN(sub)(&x3, &x3, &t);
    FN(sub)(&t, &v, &x3);
    FN(mul)(&y3, &rr, &t);
    FN(mul)(&t, &s1, &j);
    FN(dbl)(&t, &t);
    FN(sub)(&y3, &y3, &t);
    FN(add)(&z3, &p->z, &q->z);
    FN(sqr)(&z3, &z3);
    FN(sub)(&z3, &z3, &z1z1);
    FN(sub)(&z3, &z3, &z2z2);
    FN(mul)(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
/* complete mixed addition (blst_pN_add_or_double_affine) */
static inline void PF(_add_affine)(PF(_jac) *r, const PF(_jac) *p, const PF(_aff) *q) {
    PF(_jac) qj;
    PF(_from_affine)(&qj, q);
    PF(_add)(r, p, &qj);
}
/*
 * blst_pN_mult(out, P, scalar_le, nbits): [k]P for the low nbits bits of a little-endian
 * byte string; true multiplication on the whole curve (no reduction mod r, no endomorphism);
 * nbits == 0 gives infinity; out may alias P.  Fixed 4-bit windows.
 */
static inline void PF(_mult)(PF(_jac) *r, const PF(_jac) *p, const uint8_t *k_le, int nbits) {
    PF(_jac) tab[16], acc;
    PF(_set_inf)(&tab[0]);
    tab[1] = *p;
    for (int i = 2; i < 16; i++) {
        if (i & 1) PF(_add)(&tab[i], &tab[i - 1], p); else PF(_dbl)(&tab[i], &tab[i / 2]);
    }
    PF(_set_inf)(&acc);
    int top = (nbits + 3) / 4;
    for (int w = top - 1; w >= 0; w--) {
        for (int d = 0; d < 4; d++) PF(_dbl)(&acc, &acc);
        int bitpos = 4 * w, digit = 0;
        for (int b = 3; b >= 0; b--) {
            int bp = bitpos + b;
            int bit = (bp < nbits) ? (k_le[bp >> 3] >> (bp & 7)) & 1 : 0;
            digit = (digit << 1) | bit;
        }
        if (digit) PF(_add)(&acc, &acc, &tab[digit]);
    }
    *r = acc;
}
static inline int PF(_eq)(const PF(_jac) *p, const PF(_jac) *q) {
    /* equality of the represented points */
    int pi = PF(_is_inf)(p), qi = PF(_is_inf)(q);
    if (pi || qi) return pi && qi;
    F z1z1, z2z2, a, b;
    FN(sqr)(&z1z1, &p->z);
    FN(sqr)(&z2z2, &q->z);
    FN(mul)(&a, &p->x, &z2z2);
    FN(mul)(&b, &q->x, &z1z1);
    if (!FN(eq)(&a, &b)) return 0;
    FN(mul)(&a, &p->y, &q->z); FN(mul)(&a, &a, &z2z2);
    FN(mul)(&b, &q->y, &p->z); FN(mul)(&b, &b, &z1z1);
    return FN(eq)(&a, &b);
}
/* [|z|]P, |z| = 0xd201000000010000 */
static inline void PF(_mul_zabs)(PF(_jac) *r, const PF(_jac) *p) {
    uint8_t k[8];
    uint64_t z = ORA_Z_ABS;
    for (int i = 0; i < 8; i++) k[i] = (uint8_t)(z >> (8 * i));
    PF(_mult)(r, p, k, 64);
}

#undef PF
#undef EC_CAT
#undef EC_CAT_
