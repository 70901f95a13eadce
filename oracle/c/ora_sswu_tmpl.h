/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  RFC 9380 map_to_curve for one field element:
 * simplified SWU onto the isogenous curve E' (6.6.2), then the isogeny to the target curve
 * (8.8.1 / 8.8.2, appendix E).  Stands in for blst_map_to_g1 / blst_map_to_g2 with a NULL
 * second element (reference src/eip2537.c:1113,1155); cofactor clearing is done by the caller.
 * Template parameters (macros): F, FN(x), PT, SW(name) constant accessor, SGN0, SQRT, degrees.
 */
#define SW_CAT_(a, b) a##b
#define SW_CAT(a, b) SW_CAT_(a, b)
#define PF(name) SW_CAT(PT, name)

static inline void PF(_horner)(F *r, const F *coef, int ncoef, const F *x) {
    F acc = coef[ncoef - 1];
    for (int i = ncoef - 2; i >= 0; i--) {
        FN(mul)(&acc, &acc, x);
        FN(add)(&acc, &acc, &coef[i]);
    }
    *r = acc;
}

/* out = iso(sswu(u)) as an affine point of the target curve ((0,0) = infinity) */
static inline void PF(_map_to_curve)(PF(_aff) *out, const F *u) {
    const F *A = SW(A), *B = SW(B), *Zc = SW(Z);
    F u2, zu2, tv1, x1, gx1, x, y, t;
    FN(sqr)(&u2, u);
    FN(mul)(&zu2, Zc, &u2);
    FN(sqr)(&tv1, &zu2);
    FN(add)(&tv1, &tv1, &zu2);
    if (FN(is_zero)(&tv1)) {
        x1 = *SW(BZA);                               /* B / (Z A) */
    } else {
        FN(inv)(&t, &tv1);
        FN(one)(&x1);
        FN(add)(&x1, &x1, &t);
        FN(mul)(&x1, &x1, SW(MBA));                  /* (-B/A)(1 + 1/tv1) */
    }
    FN(sqr)(&gx1, &x1);
    FN(add)(&gx1, &gx1, A);
    FN(mul)(&gx1, &gx1, &x1);
    FN(add)(&gx1, &gx1, B);
    if (SQRT(&y, &gx1)) {
        x = x1;
    } else {
        F gx2;
        FN(mul)(&x, &zu2, &x1);
        FN(sqr)(&gx2, &x);
        FN(add)(&gx2, &gx2, A);
        FN(mul)(&gx2, &gx2, &x);
        FN(add)(&gx2, &gx2, B);
        SQRT(&y, &gx2);                              /* exists when gx1 is not a square */
    }
    if (SGN0(u) != SGN0(&y)) FN(neg)(&y, &y);
    /* isogeny E' -> E */
    F xn, xd, yn, yd;
    PF(_horner)(&xn, SW(XNUM), XNUM_N, &x);
    PF(_horner)(&xd, SW(XDEN), XDEN_N, &x);
    PF(_horner)(&yn, SW(YNUM), YNUM_N, &x);
    PF(_horner)(&yd, SW(YDEN), YDEN_N, &x);
    if (FN(is_zero)(&xd) || FN(is_zero)(&yd)) { memset(out, 0, sizeof *out); return; }
    FN(inv)(&t, &xd);
    FN(mul)(&out->x, &xn, &t);
    FN(inv)(&t, &yd);
    FN(mul)(&t, &t, &yn);
    FN(mul)(&out->y, &y, &t);
}

#undef PF
#undef SW_CAT
#undef SW_CAT_
