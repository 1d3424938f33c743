/*
 * ORACLE (test infrastructure, NOT product code).
 * Curve "template": included twice by groth16_oracle.c, once with the base field Fq (G1) and once
 * with Fq2 (G2).  The including file defines
 *   CT_NAME(x)   name mangler                 CT_FE      coordinate type
 *   CT_ADD/CT_SUB/CT_MUL/CT_SQR/CT_DBL(o,a[,b])  field ops   CT_ISZERO(a) CT_EQ(a,b)
 *   CT_ZERO(o) CT_ONE(o) CT_NEG(o,a) CT_INV(o,a)
 * Restates the Jacobian formulas pairing_ce's bn256 curves use (out of repo):
 * y^2 = x^3 + b, a = 0; `double`, `add_assign`, `add_assign_mixed`, `into_affine`.
 */

typedef struct { CT_FE x, y; int inf; } CT_NAME(aff);
typedef struct { CT_FE x, y, z; } CT_NAME(jac);      /* z == 0 <=> infinity */

static void CT_NAME(jac_set_inf)(CT_NAME(jac) *p) { CT_ZERO(&p->x); CT_ONE(&p->y); CT_ZERO(&p->z); }
static int CT_NAME(jac_is_inf)(const CT_NAME(jac) *p) { return CT_ISZERO(&p->z); }

static void CT_NAME(jac_from_aff)(CT_NAME(jac) *o, const CT_NAME(aff) *a) {
    if (a->inf) { CT_NAME(jac_set_inf)(o); return; }
    o->x = a->x; o->y = a->y; CT_ONE(&o->z);
}

static void CT_NAME(jac_double)(CT_NAME(jac) *p) {
    if (CT_NAME(jac_is_inf)(p)) return;
    CT_FE a, b, c, d, e, f, t;
    CT_SQR(&a, &p->x);
    CT_SQR(&b, &p->y);
    CT_SQR(&c, &b);
    CT_ADD(&d, &p->x, &b); CT_SQR(&d, &d); CT_SUB(&d, &d, &a); CT_SUB(&d, &d, &c); CT_DBL(&d, &d);
    CT_DBL(&e, &a); CT_ADD(&e, &e, &a);
    CT_SQR(&f, &e);
    CT_MUL(&p->z, &p->z, &p->y); CT_DBL(&p->z, &p->z);
    CT_SUB(&p->x, &f, &d); CT_SUB(&p->x, &p->x, &d);
    CT_SUB(&t, &d, &p->x); CT_MUL(&t, &t, &e);
    CT_DBL(&c, &c); CT_DBL(&c, &c); CT_DBL(&c, &c);
    CT_SUB(&p->y, &t, &c);
}

static void CT_NAME(jac_add)(CT_NAME(jac) *p, const CT_NAME(jac) *q) {
    if (CT_NAME(jac_is_inf)(p)) { *p = *q; return; }
    if (CT_NAME(jac_is_inf)(q)) return;
    CT_FE z1z1, z2z2, u1, u2, s1, s2, h, r, hh, hhh, v, t;
    CT_SQR(&z1z1, &p->z);
    CT_SQR(&z2z2, &q->z);
    CT_MUL(&u1, &p->x, &z2z2);
    CT_MUL(&u2, &q->x, &z1z1);
    CT_MUL(&s1, &p->y, &q->z); CT_MUL(&s1, &s1, &z2z2);
    CT_MUL(&s2, &q->y, &p->z); CT_MUL(&s2, &s2, &z1z1);
    if (CT_EQ(&u1, &u2)) {
        if (CT_EQ(&s1, &s2)) { CT_NAME(jac_double)(p); return; }
        CT_NAME(jac_set_inf)(p); return;
    }
    CT_SUB(&h, &u2, &u1);
    CT_SUB(&r, &s2, &s1);
    CT_SQR(&hh, &h);
    CT_MUL(&hhh, &hh, &h);
    CT_MUL(&v, &u1, &hh);
    CT_MUL(&p->z, &p->z, &q->z); CT_MUL(&p->z, &p->z, &h);
    CT_SQR(&p->x, &r); CT_SUB(&p->x, &p->x, &hhh); CT_SUB(&p->x, &p->x, &v); CT_SUB(&p->x, &p->x, &v);
    CT_SUB(&t, &v, &p->x); CT_MUL(&t, &t, &r);
    CT_MUL(&s1, &s1, &hhh);
    CT_SUB(&p->y, &t, &s1);
}

static void CT_NAME(jac_add_mixed)(CT_NAME(jac) *p, const CT_NAME(aff) *q) {
    if (q->inf) return;
    if (CT_NAME(jac_is_inf)(p)) { CT_NAME(jac_from_aff)(p, q); return; }
    CT_FE z1z1, u2, s2, h, r, hh, hhh, v, t, s1;
    CT_SQR(&z1z1, &p->z);
    CT_MUL(&u2, &q->x, &z1z1);
    CT_MUL(&s2, &q->y, &p->z); CT_MUL(&s2, &s2, &z1z1);
    if (CT_EQ(&p->x, &u2)) {
        if (CT_EQ(&p->y, &s2)) { CT_NAME(jac_double)(p); return; }
        CT_NAME(jac_set_inf)(p); return;
    }
    CT_SUB(&h, &u2, &p->x);
    CT_SUB(&r, &s2, &p->y);
    CT_SQR(&hh, &h);
    CT_MUL(&hhh, &hh, &h);
    CT_MUL(&v, &p->x, &hh);
    CT_MUL(&p->z, &p->z, &h);
    CT_MUL(&s1, &p->y, &hhh);
    CT_SQR(&p->x, &r); CT_SUB(&p->x, &p->x, &hhh); CT_SUB(&p->x, &p->x, &v); CT_SUB(&p->x, &p->x, &v);
    CT_SUB(&t, &v, &p->x); CT_MUL(&t, &t, &r);
    CT_SUB(&p->y, &t, &s1);
}

static void CT_NAME(jac_to_aff)(CT_NAME(aff) *o, const CT_NAME(jac) *p) {
    if (CT_NAME(jac_is_inf)(p)) { CT_ZERO(&o->x); CT_ZERO(&o->y); o->inf = 1; return; }
    CT_FE zi, zi2, zi3;
    CT_INV(&zi, &p->z);
    CT_SQR(&zi2, &zi);
    CT_MUL(&zi3, &zi2, &zi);
    CT_MUL(&o->x, &p->x, &zi2);
    CT_MUL(&o->y, &p->y, &zi3);
    o->inf = 0;
}

/* scalar given as canonical (non-Montgomery) 4x64 LE, MSB-first double-and-add */
static void CT_NAME(jac_mul)(CT_NAME(jac) *o, const CT_NAME(jac) *p, const uint64_t k[4]) {
    CT_NAME(jac) acc; CT_NAME(jac_set_inf)(&acc);
    for (int i = 255; i >= 0; i--) {
        CT_NAME(jac_double)(&acc);
        if ((k[i >> 6] >> (i & 63)) & 1) CT_NAME(jac_add)(&acc, p);
    }
    *o = acc;
}

/* bellman_ce::multiexp restated: window c = 3 if n < 32 else ceil(ln n);
 * per region: scalar 0 skipped, scalar 1 added straight to the accumulator in the first region only,
 * otherwise bucket[(s >> skip) % 2^c - 1] += base (mixed add); summation by parts; regions joined
 * high -> low with c doublings.  `density` (may be NULL = FullDensity) selects which scalars take part;
 * bases are consumed one per *selected* scalar (the key arrays are compacted).
 * exps: canonical 4x64 LE per scalar. */
static unsigned CT_NAME(multiexp_window)(size_t n_exps) {      /* bellman sizes c from exponents.len() (the un-filtered length) */
    return n_exps < 32 ? 3 : (unsigned)ceil(log((double)(uint32_t)n_exps));
}
static unsigned CT_NAME(multiexp_regions)(unsigned c) {
    unsigned nregions = 0;
    for (unsigned skip = 0; ; skip += c) { nregions++; if (skip + c >= 254) break; }
    return nregions;
}
/* one region (window) of the sum: bellman spawns exactly this unit of work on its thread pool (multiexp_inner) */
static void CT_NAME(multiexp_region)(CT_NAME(jac) *out, const CT_NAME(aff) *bases, const uint8_t *density,
                                     const uint64_t *exps, size_t n_exps, unsigned c, unsigned reg) {
    const size_t nb = ((size_t)1 << c) - 1;
    const unsigned skip = reg * c;
    CT_NAME(jac) *buckets = (CT_NAME(jac) *)malloc(nb * sizeof(CT_NAME(jac)));
    CT_NAME(jac) acc; CT_NAME(jac_set_inf)(&acc);
    for (size_t i = 0; i < nb; i++) CT_NAME(jac_set_inf)(&buckets[i]);
    size_t bi = 0;
    for (size_t i = 0; i < n_exps; i++) {
        if (density && !density[i]) continue;
        const uint64_t *e = exps + 4 * i;
        const CT_NAME(aff) *base = &bases[bi++];
        if ((e[0] | e[1] | e[2] | e[3]) == 0) continue;
        if (e[0] == 1 && (e[1] | e[2] | e[3]) == 0) {
            if (reg == 0) CT_NAME(jac_add_mixed)(&acc, base);
            continue;
        }
        /* (e >> skip) mod 2^c */
        unsigned limb = skip >> 6, off = skip & 63;
        uint64_t v = e[limb] >> off;
        if (off && limb + 1 < 4) v |= e[limb + 1] << (64 - off);
        v &= ((uint64_t)1 << c) - 1;
        if (v) CT_NAME(jac_add_mixed)(&buckets[v - 1], base);
    }
    CT_NAME(jac) running; CT_NAME(jac_set_inf)(&running);
    for (size_t i = nb; i-- > 0;) {
        CT_NAME(jac_add)(&running, &buckets[i]);
        CT_NAME(jac_add)(&acc, &running);
    }
    *out = acc;
    free(buckets);
}
/* regions joined high -> low with c doublings */
static void CT_NAME(multiexp_join)(CT_NAME(jac) *out, const CT_NAME(jac) *regions, unsigned nregions, unsigned c) {
    CT_NAME(jac) res = regions[nregions - 1];
    for (unsigned reg = nregions - 1; reg-- > 0;) {
        for (unsigned k = 0; k < c; k++) CT_NAME(jac_double)(&res);
        CT_NAME(jac_add)(&res, &regions[reg]);
    }
    *out = res;
}
static void CT_NAME(multiexp)(CT_NAME(jac) *out, const CT_NAME(aff) *bases, const uint8_t *density,
                              const uint64_t *exps, size_t n_exps) {
    const unsigned c = CT_NAME(multiexp_window)(n_exps), nregions = CT_NAME(multiexp_regions)(c);
    CT_NAME(jac) *regions = (CT_NAME(jac) *)malloc(nregions * sizeof(CT_NAME(jac)));
    /* one task per region, as bellman's multiexp_inner spawns them (ORC_THREADS = 1: the serial order; same result either way) */
    #pragma omp parallel for schedule(dynamic, 1) num_threads(ORC_THREADS)
    for (unsigned reg = 0; reg < nregions; reg++) CT_NAME(multiexp_region)(&regions[reg], bases, density, exps, n_exps, c, reg);
    CT_NAME(multiexp_join)(out, regions, nregions, c);
    free(regions);
}

/* fixed-base table for key generation: tbl[w][d] = d * 2^(8w) * G, d in 1..255 (affine).  bellman
 * uses a wNAF table; any correct fixed-base method yields the same points. */
typedef struct { CT_NAME(aff) *t; } CT_NAME(fbtable);

static void CT_NAME(fb_init)(CT_NAME(fbtable) *T, const CT_NAME(aff) *g) {
    T->t = (CT_NAME(aff) *)malloc(32 * 255 * sizeof(CT_NAME(aff)));
    CT_NAME(jac) base; CT_NAME(jac_from_aff)(&base, g);
    for (int w = 0; w < 32; w++) {
        CT_NAME(jac) cur = base;
        for (int d = 1; d <= 255; d++) {
            CT_NAME(jac_to_aff)(&T->t[w * 255 + d - 1], &cur);
            CT_NAME(jac_add)(&cur, &base);
        }
        base = cur; /* 256 * previous base */
    }
}

static void CT_NAME(fb_mul)(CT_NAME(aff) *o, const CT_NAME(fbtable) *T, const uint64_t k[4]) {
    CT_NAME(jac) acc; CT_NAME(jac_set_inf)(&acc);
    for (int w = 0; w < 32; w++) {
        unsigned d = (unsigned)((k[w >> 3] >> ((w & 7) * 8)) & 0xff);
        if (d) CT_NAME(jac_add_mixed)(&acc, &T->t[w * 255 + d - 1]);
    }
    CT_NAME(jac_to_aff)(o, &acc);
}
