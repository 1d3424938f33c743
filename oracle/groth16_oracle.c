/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the CPU algorithm behind
 * fawkes-crypto's `backend::bellman_groth16::prover::prove`
 * (/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:63-90).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and only as the checker / the timed CPU baseline -- never as a fallback for the HIP path.
 *
 * PARITY STATUS: **parity unpinned**.  The heavy call, `bellman::groth16::create_random_proof`
 * (prover.rs:80), lives in the crates.io packages fawkes-crypto-bellman_ce 0.3.5 /
 * fawkes-crypto-pairing_ce 0.18.1 / ff_ce 0.7.1 (Cargo.lock:413-436,495-504), which are NOT
 * under /root/reference, and no Rust toolchain exists in this image, so the reference cannot be
 * built or run here (unbuildable; no oracle/_ref).  This file restates bellman_ce's published
 * algorithm: `domain.rs` serial radix-2 FFT + coset recipe, `multiexp.rs`
 * Pippenger with its window rule, `groth16/prover.rs` proof assembly, `groth16/generator.rs` key
 * generation.  It is anchored on what the reference itself holds:
 *   - Montgomery field convention of ff-uint (R = 2^256, INV = -p^-1 mod 2^64;
 *     ff-uint_derive/src/lib.rs:229-253,354-366, reduction :434-490, mul :578-623), checked
 *     against the reference's own known-answer tests ff-uint/tests/ff-uint_tests.rs:35-156
 *     (tests/golden/ff_uint_kats.json) through the run-time modulus of `orc_field_custom`;
 *   - moduli and generator: fawkes-crypto/src/engines/bn256/mod.rs:13,23,24;
 *   - byte layouts: group.rs:54-80,88-122 (raw Montgomery LE points, zero = infinity),
 *     prover.rs:39-60 (256-byte Borsh proof);
 *   - witness/row order: circuit/r1cs/cs.rs:255-268, backend/bellman_groth16/mod.rs:61-102;
 * and on the Groth16 pairing equation checked by oracle/bn254_ref.py (an independent big-int
 * implementation), plus Python<->C agreement on tests/golden/ vectors.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic 4-limb Montgomery field */
typedef struct { uint64_t l[4]; } fe;
typedef struct {
    uint64_t p[4];
    uint64_t inv;      /* -p^-1 mod 2^64 */
    fe r, r2;          /* R mod p, R^2 mod p */
    uint64_t pm2[4];   /* p - 2 */
} field_t;

static field_t FQ, FR, FX;

static int u256_geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static uint64_t u256_sub(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - br; o[i] = (uint64_t)t; br = (t >> 64) & 1; }
    return (uint64_t)br;
}
static uint64_t u256_add(uint64_t o[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}

static void fe_add(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[4]; uint64_t c = u256_add(t, a->l, b->l);
    if (c || u256_geq(t, F->p)) u256_sub(t, t, F->p);
    memcpy(o->l, t, 32);
}
static void fe_sub(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[4]; uint64_t br = u256_sub(t, a->l, b->l);
    if (br) u256_add(t, t, F->p);
    memcpy(o->l, t, 32);
}
static void fe_neg(const field_t *F, fe *o, const fe *a) {
    if ((a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0) { memset(o, 0, 32); return; }
    uint64_t t[4]; u256_sub(t, F->p, a->l); memcpy(o->l, t, 32);
}
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, 32) == 0; }

/* CIOS Montgomery product a*b*R^-1 mod p */
static void fe_mul(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || u256_geq(t, F->p)) u256_sub(t, t, F->p);
    memcpy(o->l, t, 32);
}
static void fe_sqr(const field_t *F, fe *o, const fe *a) { fe_mul(F, o, a, a); }
static void fe_pow(const field_t *F, fe *o, const fe *a, const uint64_t e[4]) {
    fe acc = F->r, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, &base);
        fe_mul(F, &base, &base, &base);
    }
    *o = acc;
}
static void fe_inv(const field_t *F, fe *o, const fe *a) { fe_pow(F, o, a, F->pm2); }
static void fe_from_canon(const field_t *F, fe *o, const uint64_t c[4]) {
    fe t; memcpy(t.l, c, 32);
    while (u256_geq(t.l, F->p)) u256_sub(t.l, t.l, F->p);
    fe_mul(F, o, &t, &F->r2);
}
static void fe_to_canon(const field_t *F, uint64_t c[4], const fe *a) {
    fe one = {{1, 0, 0, 0}}, t; fe_mul(F, &t, a, &one); memcpy(c, t.l, 32);
}
static void fe_from_u64(const field_t *F, fe *o, uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; fe_from_canon(F, o, c); }

static void field_init(field_t *F, const uint64_t p[4]) {
    memcpy(F->p, p, 32);
    uint64_t inv = 1;
    for (int i = 0; i < 63; i++) { inv = inv * inv; inv = inv * p[0]; }  /* p^(2^63-1) = p^-1 mod 2^64 */
    F->inv = (uint64_t)(0 - inv);
    /* R mod p by 256 modular doublings of 1; R2 by 256 more */
    fe x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        uint64_t t[4]; uint64_t c = u256_add(t, x.l, x.l);
        if (c || u256_geq(t, p)) u256_sub(t, t, p);
        memcpy(x.l, t, 32);
        if (i == 255) F->r = x;
    }
    F->r2 = x;
    uint64_t two[4] = {2, 0, 0, 0};
    u256_sub(F->pm2, p, two);
}

static const uint64_t BN_Q[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t BN_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#define FR_S 28

static int g_inited = 0;
static fe FR_ROOT;     /* 7^((r-1)/2^28), Montgomery */
static fe FR_GEN;      /* 7 */

/* ------------------------------------------------------------------ Fq / Fq2 wrappers for the curve template */
typedef struct { fe c0, c1; } fe2;
static void q_add(fe *o, const fe *a, const fe *b) { fe_add(&FQ, o, a, b); }
static void q_sub(fe *o, const fe *a, const fe *b) { fe_sub(&FQ, o, a, b); }
static void q_mul(fe *o, const fe *a, const fe *b) { fe_mul(&FQ, o, a, b); }
static void q_sqr(fe *o, const fe *a) { fe_mul(&FQ, o, a, a); }
static void q_dbl(fe *o, const fe *a) { fe_add(&FQ, o, a, a); }
static void q_neg(fe *o, const fe *a) { fe_neg(&FQ, o, a); }
static void q_inv(fe *o, const fe *a) { fe_inv(&FQ, o, a); }
static void q_zero(fe *o) { memset(o, 0, sizeof *o); }
static void q_one(fe *o) { *o = FQ.r; }

static void q2_add(fe2 *o, const fe2 *a, const fe2 *b) { q_add(&o->c0, &a->c0, &b->c0); q_add(&o->c1, &a->c1, &b->c1); }
static void q2_sub(fe2 *o, const fe2 *a, const fe2 *b) { q_sub(&o->c0, &a->c0, &b->c0); q_sub(&o->c1, &a->c1, &b->c1); }
static void q2_dbl(fe2 *o, const fe2 *a) { q2_add(o, a, a); }
static void q2_neg(fe2 *o, const fe2 *a) { q_neg(&o->c0, &a->c0); q_neg(&o->c1, &a->c1); }
static void q2_mul(fe2 *o, const fe2 *a, const fe2 *b) {  /* u^2 = -1 */
    fe aa, bb, t0, t1;
    q_mul(&aa, &a->c0, &b->c0);
    q_mul(&bb, &a->c1, &b->c1);
    q_add(&t0, &a->c0, &a->c1);
    q_add(&t1, &b->c0, &b->c1);
    q_mul(&t0, &t0, &t1);
    q_sub(&t0, &t0, &aa); q_sub(&t0, &t0, &bb);
    q_sub(&o->c0, &aa, &bb);
    o->c1 = t0;
}
static void q2_sqr(fe2 *o, const fe2 *a) { q2_mul(o, a, a); }
static void q2_inv(fe2 *o, const fe2 *a) {
    fe n, t; q_sqr(&n, &a->c0); q_sqr(&t, &a->c1); q_add(&n, &n, &t); q_inv(&n, &n);
    q_mul(&o->c0, &a->c0, &n); q_mul(&t, &a->c1, &n); q_neg(&o->c1, &t);
}
static void q2_zero(fe2 *o) { memset(o, 0, sizeof *o); }
static void q2_one(fe2 *o) { o->c0 = FQ.r; memset(&o->c1, 0, 32); }
static int q2_is_zero(const fe2 *a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static int q2_eq(const fe2 *a, const fe2 *b) { return memcmp(a, b, 64) == 0; }

#define CT_NAME(x) g1_##x
#define CT_FE fe
#define CT_ADD q_add
#define CT_SUB q_sub
#define CT_MUL q_mul
#define CT_SQR q_sqr
#define CT_DBL q_dbl
#define CT_NEG q_neg
#define CT_INV q_inv
#define CT_ZERO q_zero
#define CT_ONE q_one
#define CT_ISZERO fe_is_zero
#define CT_EQ fe_eq
static int ORC_THREADS;      /* worker threads (defined below with orc_set_threads); the multiexp template reads it */
#include "curve_tmpl.h"
#undef CT_NAME
#undef CT_FE
#undef CT_ADD
#undef CT_SUB
#undef CT_MUL
#undef CT_SQR
#undef CT_DBL
#undef CT_NEG
#undef CT_INV
#undef CT_ZERO
#undef CT_ONE
#undef CT_ISZERO
#undef CT_EQ

#define CT_NAME(x) g2_##x
#define CT_FE fe2
#define CT_ADD q2_add
#define CT_SUB q2_sub
#define CT_MUL q2_mul
#define CT_SQR q2_sqr
#define CT_DBL q2_dbl
#define CT_NEG q2_neg
#define CT_INV q2_inv
#define CT_ZERO q2_zero
#define CT_ONE q2_one
#define CT_ISZERO q2_is_zero
#define CT_EQ q2_eq
#include "curve_tmpl.h"

/* raw Montgomery-LE point buffers (group.rs:57-66, :97-103); all-zero = infinity (group.rs:55) */
static void g1_load(g1_aff *o, const uint8_t *b) {
    memcpy(&o->x, b, 32); memcpy(&o->y, b + 32, 32);
    o->inf = fe_is_zero(&o->x) && fe_is_zero(&o->y);
}
static void g1_store(uint8_t *b, const g1_aff *a) {
    if (a->inf) { memset(b, 0, 64); return; }
    memcpy(b, &a->x, 32); memcpy(b + 32, &a->y, 32);
}
static void g2_load(g2_aff *o, const uint8_t *b) {
    memcpy(&o->x, b, 64); memcpy(&o->y, b + 64, 64);
    o->inf = q2_is_zero(&o->x) && q2_is_zero(&o->y);
}
static void g2_store(uint8_t *b, const g2_aff *a) {
    if (a->inf) { memset(b, 0, 128); return; }
    memcpy(b, &a->x, 64); memcpy(b + 64, &a->y, 64);
}

/* ------------------------------------------------------------------ init */
void orc_init(void) {
    if (g_inited) return;
    field_init(&FQ, BN_Q);
    field_init(&FR, BN_R);
    fe_from_u64(&FR, &FR_GEN, 7);
    /* t = (r-1) >> 28 */
    uint64_t e[4]; uint64_t one[4] = {1, 0, 0, 0};
    u256_sub(e, BN_R, one);
    for (int i = 0; i < FR_S; i++) {
        for (int k = 0; k < 3; k++) e[k] = (e[k] >> 1) | (e[k + 1] << 63);
        e[3] >>= 1;
    }
    fe_pow(&FR, &FR_ROOT, &FR_GEN, e);
    g_inited = 1;
}

static const field_t *field_by_id(int id) { return id == 0 ? &FQ : (id == 1 ? &FR : &FX); }

/* ------------------------------------------------------------------ exported field micro-API */
void orc_field_custom(const uint64_t modulus[4]) { field_init(&FX, modulus); }
void orc_fe_from_canon(int f, const uint64_t in[4], uint64_t out[4]) { orc_init(); fe t; fe_from_canon(field_by_id(f), &t, in); memcpy(out, t.l, 32); }
void orc_fe_to_canon(int f, const uint64_t in[4], uint64_t out[4]) { orc_init(); fe t; memcpy(t.l, in, 32); fe_to_canon(field_by_id(f), out, &t); }
void orc_fe_add(int f, const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { orc_init(); fe_add(field_by_id(f), (fe *)o, (const fe *)a, (const fe *)b); }
void orc_fe_sub(int f, const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { orc_init(); fe_sub(field_by_id(f), (fe *)o, (const fe *)a, (const fe *)b); }
void orc_fe_mul(int f, const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { orc_init(); fe_mul(field_by_id(f), (fe *)o, (const fe *)a, (const fe *)b); }
void orc_fe_neg(int f, const uint64_t a[4], uint64_t o[4]) { orc_init(); fe_neg(field_by_id(f), (fe *)o, (const fe *)a); }
void orc_fe_inv(int f, const uint64_t a[4], uint64_t o[4]) { orc_init(); fe_inv(field_by_id(f), (fe *)o, (const fe *)a); }
void orc_fe_pow(int f, const uint64_t a[4], const uint64_t e[4], uint64_t o[4]) { orc_init(); fe_pow(field_by_id(f), (fe *)o, (const fe *)a, e); }
void orc_fe_mul_batch(int f, const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    orc_init(); const field_t *F = field_by_id(f);
    for (size_t i = 0; i < n; i++) fe_mul(F, (fe *)(o + 4 * i), (const fe *)(a + 4 * i), (const fe *)(b + 4 * i));
}

/* ------------------------------------------------------------------ bellman_ce::domain restated */
static uint32_t bitrev(uint32_t n, uint32_t l) { uint32_t r = 0; for (uint32_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; } return r; }

static void fr_pow_u64(fe *o, const fe *a, uint64_t e) { uint64_t ee[4] = {e, 0, 0, 0}; fe_pow(&FR, o, a, ee); }

static void serial_fft(fe *a, const fe *omega, uint32_t log_n) {
    uint32_t n = 1u << log_n;
    for (uint32_t k = 0; k < n; k++) { uint32_t rk = bitrev(k, log_n); if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; } }
    uint32_t m = 1;
    for (uint32_t s = 0; s < log_n; s++) {
        fe w_m; fr_pow_u64(&w_m, omega, n / (2 * m));
        for (uint32_t k = 0; k < n; k += 2 * m) {
            fe w = FR.r;
            for (uint32_t j = 0; j < m; j++) {
                fe t; fe_mul(&FR, &t, &a[k + j + m], &w);
                fe tmp; fe_sub(&FR, &tmp, &a[k + j], &t);
                a[k + j + m] = tmp;
                fe_add(&FR, &a[k + j], &a[k + j], &t);
                fe_mul(&FR, &w, &w, &w_m);
            }
        }
        m *= 2;
    }
}

/* Worker threads of the prover (bellman's `Worker`).  1 = the serial path, which is also what fawkes-crypto configures
 * (single-core worker); > 1 = bellman's multicore split, restated below: the FFT is cut into
 * 2^log_cpus sub-FFTs (domain.rs: parallel_fft), the pointwise passes into chunks, and every multiexp region is one task
 * (multiexp.rs: multiexp_inner spawns a task per region).  All of it is exact field / group arithmetic: the proof bytes do
 * not depend on the thread count (tests/test_oracle.py). */
static int ORC_THREADS = 1;
void orc_set_threads(int t) {
#ifdef _OPENMP
    ORC_THREADS = t < 1 ? 1 : t;
#else
    (void)t; ORC_THREADS = 1;
#endif
}
static uint32_t log2_floor_u32(uint32_t v) { uint32_t l = 0; while ((2u << l) <= v) l++; return l; }

/* bellman_ce domain.rs: parallel_fft.  2^log_cpus sub-FFTs of size n / 2^log_cpus: sub-FFT j collects
 * tmp[i] = sum_s a[(i + s * new_n) mod n] * omega^(j * (i + s * new_n)), transforms with omega^num_cpus, and the results
 * are interleaved back: a[idx] = tmp[idx mod num_cpus][idx / num_cpus]. */
static void parallel_fft(fe *a, const fe *omega, uint32_t log_n, uint32_t log_cpus) {
    const uint32_t num_cpus = 1u << log_cpus, log_new_n = log_n - log_cpus;
    const uint64_t new_n = (uint64_t)1 << log_new_n, n = (uint64_t)1 << log_n;
    fe *tmp = (fe *)calloc(n, sizeof(fe));
    fe new_omega; fr_pow_u64(&new_omega, omega, num_cpus);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(ORC_THREADS)
    for (uint32_t j = 0; j < num_cpus; j++) {
        fe *t = tmp + (uint64_t)j * new_n;
        fe omega_j, omega_step, elt = FR.r;
        fr_pow_u64(&omega_j, omega, j);
        fr_pow_u64(&omega_step, omega, (uint64_t)j << log_new_n);
        for (uint64_t i = 0; i < new_n; i++) {
            for (uint32_t s_ = 0; s_ < num_cpus; s_++) {
                const uint64_t idx = (i + ((uint64_t)s_ << log_new_n)) & (n - 1);
                fe x; fe_mul(&FR, &x, &a[idx], &elt);
                fe_add(&FR, &t[i], &t[i], &x);
                fe_mul(&FR, &elt, &elt, &omega_step);
            }
            fe_mul(&FR, &elt, &elt, &omega_j);
        }
        serial_fft(t, &new_omega, log_new_n);
    }
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (uint64_t idx = 0; idx < n; idx++) a[idx] = tmp[(idx & (num_cpus - 1)) * new_n + (idx >> log_cpus)];
    free(tmp);
}
/* domain.rs: best_fft -- serial when the transform is no larger than the core count */
static void best_fft(fe *a, const fe *omega, uint32_t log_n) {
    const uint32_t log_cpus = log2_floor_u32((uint32_t)ORC_THREADS);
    if (ORC_THREADS <= 1 || log_n <= log_cpus) serial_fft(a, omega, log_n);
    else parallel_fft(a, omega, log_n, log_cpus);
}

typedef struct { uint32_t exp; uint64_t m; fe omega, omegainv, geninv, minv; } domain_t;

/* EvaluationDomain::from_coeffs: m = 2^exp >= n; error once exp >= S (PolynomialDegreeTooLarge) */
static int domain_init(domain_t *d, uint64_t n) {
    uint64_t m = 1; uint32_t exp = 0;
    while (m < n) { m *= 2; exp += 1; if (exp >= FR_S) return -1; }
    d->exp = exp; d->m = m;
    d->omega = FR_ROOT;
    for (uint32_t i = exp; i < FR_S; i++) fe_mul(&FR, &d->omega, &d->omega, &d->omega);
    fe_inv(&FR, &d->omegainv, &d->omega);
    fe_inv(&FR, &d->geninv, &FR_GEN);
    fe mm; fe_from_u64(&FR, &mm, m); fe_inv(&FR, &d->minv, &mm);
    return 0;
}
static void dom_fft(const domain_t *d, fe *a) { best_fft(a, &d->omega, d->exp); }
static void dom_ifft(const domain_t *d, fe *a) {
    best_fft(a, &d->omegainv, d->exp);
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (uint64_t i = 0; i < d->m; i++) fe_mul(&FR, &a[i], &a[i], &d->minv);
}
/* domain.rs: distribute_powers -- every worker chunk starts from g^(first index of the chunk) */
static void dom_distribute_powers(const domain_t *d, fe *a, const fe *g) {
    const uint64_t nch = ORC_THREADS > 1 ? (uint64_t)ORC_THREADS : 1, chunk = (d->m + nch - 1) / nch;
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (uint64_t ch = 0; ch < nch; ch++) {
        const uint64_t lo = ch * chunk, hi = lo + chunk < d->m ? lo + chunk : d->m;
        if (lo >= hi) continue;
        fe u; fr_pow_u64(&u, g, lo);
        for (uint64_t i = lo; i < hi; i++) { fe_mul(&FR, &a[i], &a[i], &u); fe_mul(&FR, &u, &u, g); }
    }
}
static void dom_coset_fft(const domain_t *d, fe *a) { dom_distribute_powers(d, a, &FR_GEN); dom_fft(d, a); }
static void dom_icoset_fft(const domain_t *d, fe *a) { dom_ifft(d, a); dom_distribute_powers(d, a, &d->geninv); }
static void dom_divide_by_z_on_coset(const domain_t *d, fe *a) {
    fe i; fr_pow_u64(&i, &FR_GEN, d->m); fe_sub(&FR, &i, &i, &FR.r); fe_inv(&FR, &i, &i);
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (uint64_t k = 0; k < d->m; k++) fe_mul(&FR, &a[k], &a[k], &i);
}

/* natural-order NTT of 2^log_n Montgomery Fr elements, in place.  inverse != 0: omega^-1 and 1/n. */
int orc_fr_ntt(uint64_t *a, uint32_t log_n, int inverse) {
    orc_init(); domain_t d; if (domain_init(&d, (uint64_t)1 << log_n)) return -1;
    if (inverse) dom_ifft(&d, (fe *)a); else dom_fft(&d, (fe *)a);
    return 0;
}
int orc_fr_coset_ntt(uint64_t *a, uint32_t log_n, int inverse) {
    orc_init(); domain_t d; if (domain_init(&d, (uint64_t)1 << log_n)) return -1;
    if (inverse) dom_icoset_fft(&d, (fe *)a); else dom_coset_fft(&d, (fe *)a);
    return 0;
}

/* h = (A*B-C)/Z: 3 ifft + 3 coset_fft, pointwise, divide_by_z_on_coset, icoset_fft, drop last coeff.
 * a,b,c: n rows each (Montgomery); out: m-1 coefficients (Montgomery).  Returns m or 0 on error. */
uint64_t orc_quotient_h(const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n, uint64_t *h_out) {
    orc_init(); domain_t d; if (domain_init(&d, n)) return 0;
    fe *A = (fe *)calloc(d.m, sizeof(fe)), *B = (fe *)calloc(d.m, sizeof(fe)), *C = (fe *)calloc(d.m, sizeof(fe));
    memcpy(A, a, n * 32); memcpy(B, b, n * 32); memcpy(C, c, n * 32);
    dom_ifft(&d, A); dom_coset_fft(&d, A);
    dom_ifft(&d, B); dom_coset_fft(&d, B);
    dom_ifft(&d, C); dom_coset_fft(&d, C);
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (uint64_t i = 0; i < d.m; i++) { fe_mul(&FR, &A[i], &A[i], &B[i]); fe_sub(&FR, &A[i], &A[i], &C[i]); }
    dom_divide_by_z_on_coset(&d, A);
    dom_icoset_fft(&d, A);
    memcpy(h_out, A, (d.m - 1) * 32);
    free(A); free(B); free(C);
    return d.m;
}

/* ------------------------------------------------------------------ MSM entry points */
static uint64_t *scalars_to_canon(const uint64_t *mont, size_t n) {
    uint64_t *c = (uint64_t *)malloc(n * 32 + 32);
    #pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (size_t i = 0; i < n; i++) fe_to_canon(&FR, c + 4 * i, (const fe *)(mont + 4 * i));
    return c;
}

/* bases: n x 64 B raw LE; scalars: n x 4 u64 Montgomery Fr (as they sit in WitnessCS); density: NULL or
 * one byte per scalar (bases then hold popcount(density) points).  out: 64 B raw LE affine. */
void orc_msm_g1(const uint8_t *bases, size_t n_bases, const uint64_t *scalars, const uint8_t *density, size_t n_scalars, uint8_t out[64]) {
    orc_init();
    g1_aff *B = (g1_aff *)malloc((n_bases + 1) * sizeof(g1_aff));
    for (size_t i = 0; i < n_bases; i++) g1_load(&B[i], bases + 64 * i);
    uint64_t *e = scalars_to_canon(scalars, n_scalars);
    g1_jac r; g1_multiexp(&r, B, density, e, n_scalars);
    g1_aff ra; g1_jac_to_aff(&ra, &r); g1_store(out, &ra);
    free(B); free(e);
}
void orc_msm_g2(const uint8_t *bases, size_t n_bases, const uint64_t *scalars, const uint8_t *density, size_t n_scalars, uint8_t out[128]) {
    orc_init();
    g2_aff *B = (g2_aff *)malloc((n_bases + 1) * sizeof(g2_aff));
    for (size_t i = 0; i < n_bases; i++) g2_load(&B[i], bases + 128 * i);
    uint64_t *e = scalars_to_canon(scalars, n_scalars);
    g2_jac r; g2_multiexp(&r, B, density, e, n_scalars);
    g2_aff ra; g2_jac_to_aff(&ra, &r); g2_store(out, &ra);
    free(B); free(e);
}

/* k*P helpers for tests (k Montgomery Fr) */
void orc_g1_mul(const uint8_t p[64], const uint64_t k_mont[4], uint8_t out[64]) {
    orc_init(); g1_aff a; g1_load(&a, p); g1_jac j; g1_jac_from_aff(&j, &a);
    uint64_t k[4]; fe_to_canon(&FR, k, (const fe *)k_mont);
    g1_jac_mul(&j, &j, k); g1_jac_to_aff(&a, &j); g1_store(out, &a);
}
void orc_g2_mul(const uint8_t p[128], const uint64_t k_mont[4], uint8_t out[128]) {
    orc_init(); g2_aff a; g2_load(&a, p); g2_jac j; g2_jac_from_aff(&j, &a);
    uint64_t k[4]; fe_to_canon(&FR, k, (const fe *)k_mont);
    g2_jac_mul(&j, &j, k); g2_jac_to_aff(&a, &j); g2_store(out, &a);
}
void orc_g1_add(const uint8_t p[64], const uint8_t q[64], uint8_t out[64]) {
    orc_init(); g1_aff a, b; g1_load(&a, p); g1_load(&b, q); g1_jac j; g1_jac_from_aff(&j, &a);
    g1_jac_add_mixed(&j, &b); g1_jac_to_aff(&a, &j); g1_store(out, &a);
}
void orc_g2_add(const uint8_t p[128], const uint8_t q[128], uint8_t out[128]) {
    orc_init(); g2_aff a, b; g2_load(&a, p); g2_load(&b, q); g2_jac j; g2_jac_from_aff(&j, &a);
    g2_jac_add_mixed(&j, &b); g2_jac_to_aff(&a, &j); g2_store(out, &a);
}
/* n points i*step*G + start*G ... used to make synthetic key material quickly: out[i] = (start + i) * P */
void orc_g1_series(const uint8_t p[64], const uint64_t start_mont[4], size_t n, uint8_t *out) {
    orc_init(); g1_aff a; g1_load(&a, p); g1_jac cur; g1_jac_from_aff(&cur, &a);
    uint64_t k[4]; fe_to_canon(&FR, k, (const fe *)start_mont);
    g1_jac_mul(&cur, &cur, k);
    for (size_t i = 0; i < n; i++) { g1_aff t; g1_jac_to_aff(&t, &cur); g1_store(out + 64 * i, &t); g1_jac_add_mixed(&cur, &a); }
}
void orc_g2_series(const uint8_t p[128], const uint64_t start_mont[4], size_t n, uint8_t *out) {
    orc_init(); g2_aff a; g2_load(&a, p); g2_jac cur; g2_jac_from_aff(&cur, &a);
    uint64_t k[4]; fe_to_canon(&FR, k, (const fe *)start_mont);
    g2_jac_mul(&cur, &cur, k);
    for (size_t i = 0; i < n; i++) { g2_aff t; g2_jac_to_aff(&t, &cur); g2_store(out + 128 * i, &t); g2_jac_add_mixed(&cur, &a); }
}

/* ------------------------------------------------------------------ R1CS in CSR form
 * Variables are numbered Input(i) -> i, Aux(j) -> num_input + j (cs.rs:255-268; Input(0) = ONE).
 * Each matrix: row_ptr[num_gates+1], col[nnz], coeff[nnz] (Montgomery Fr, 4 x u64). */
typedef struct {
    uint32_t num_input, num_aux; uint64_t num_gates;
    const uint64_t *a_ptr; const uint32_t *a_col; const uint64_t *a_val;
    const uint64_t *b_ptr; const uint32_t *b_col; const uint64_t *b_val;
    const uint64_t *c_ptr; const uint32_t *c_col; const uint64_t *c_val;
} orc_r1cs;

/* ProvingAssignment::enforce / eval restated (Appendix A.1).  z = z_input || z_aux (Montgomery).
 * Outputs: a,b,c with n = num_gates + num_input rows; density bytes (0/1): a_aux[num_aux],
 * b_input[num_input], b_aux[num_aux]. */
static void eval_lc(fe *out, const uint64_t *ptr, const uint32_t *col, const uint64_t *val, uint64_t row,
                    const fe *z, uint32_t num_input, uint8_t *din, uint8_t *daux) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (uint64_t k = ptr[row]; k < ptr[row + 1]; k++) {
        uint32_t v = col[k];
        if (v < num_input) { if (din) din[v] = 1; } else { if (daux) daux[v - num_input] = 1; }
        fe t = z[v];
        const fe *cf = (const fe *)(val + 4 * k);
        if (!fe_eq(cf, &FR.r)) fe_mul(&FR, &t, &t, cf);
        fe_add(&FR, &acc, &acc, &t);
    }
    *out = acc;
}

void orc_synthesize(const orc_r1cs *cs, const uint64_t *z, uint64_t *a, uint64_t *b, uint64_t *c,
                    uint8_t *a_aux, uint8_t *b_input, uint8_t *b_aux) {
    orc_init();
    memset(a_aux, 0, cs->num_aux); memset(b_input, 0, cs->num_input); memset(b_aux, 0, cs->num_aux);
    const fe *Z = (const fe *)z;
    for (uint64_t g = 0; g < cs->num_gates; g++) {
        eval_lc((fe *)(a + 4 * g), cs->a_ptr, cs->a_col, cs->a_val, g, Z, cs->num_input, NULL, a_aux);
        eval_lc((fe *)(b + 4 * g), cs->b_ptr, cs->b_col, cs->b_val, g, Z, cs->num_input, b_input, b_aux);
        eval_lc((fe *)(c + 4 * g), cs->c_ptr, cs->c_col, cs->c_val, g, Z, cs->num_input, NULL, NULL);
    }
    for (uint32_t i = 0; i < cs->num_input; i++) {   /* input_i * 0 = 0 */
        uint64_t row = cs->num_gates + i;
        memcpy(a + 4 * row, Z + i, 32); memset(b + 4 * row, 0, 32); memset(c + 4 * row, 0, 32);
    }
}

/* The same evaluation for a batch circuit given as ONE instance + a copy count (the system fixtures.tile_r1cs writes out
 * explicitly: rows of copy j are [j*G, (j+1)*G); variables ONE, copy 0's inputs, copy 1's inputs, ..., copy 0's aux, ...),
 * without materialising copies x nnz terms -- what lets the CPU baseline run at the benchmark's full size (9.6e8 terms).
 * Same order of additions per row as orc_synthesize on the replicated system, hence the same field elements. */
static void eval_lc_tiled(fe *out, const uint64_t *ptr, const uint32_t *col, const uint64_t *val, uint64_t row, const fe *z,
                          uint32_t base_input, uint32_t in_off, uint32_t aux_off, uint32_t num_input, uint8_t *din, uint8_t *daux) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (uint64_t k = ptr[row]; k < ptr[row + 1]; k++) {
        uint32_t v = col[k];
        if (v) v += v < base_input ? in_off : aux_off;
        if (v < num_input) { if (din) din[v] = 1; } else { if (daux) daux[v - num_input] = 1; }
        fe t = z[v];
        const fe *cf = (const fe *)(val + 4 * k);
        if (!fe_eq(cf, &FR.r)) fe_mul(&FR, &t, &t, cf);
        fe_add(&FR, &acc, &acc, &t);
    }
    *out = acc;
}

void orc_synthesize_tiled(const orc_r1cs *inst, uint32_t copies, const uint64_t *z, uint64_t *a, uint64_t *b, uint64_t *c,
                          uint8_t *a_aux, uint8_t *b_input, uint8_t *b_aux) {
    orc_init();
    const uint32_t num_input = 1 + copies * (inst->num_input - 1), num_aux = copies * inst->num_aux;
    const uint64_t G = inst->num_gates;
    memset(a_aux, 0, num_aux); memset(b_input, 0, num_input); memset(b_aux, 0, num_aux);
    const fe *Z = (const fe *)z;
    for (uint32_t j = 0; j < copies; j++) {
        const uint32_t in_off = j * (inst->num_input - 1), aux_off = num_input + j * inst->num_aux - inst->num_input;
        for (uint64_t g = 0; g < G; g++) {
            const uint64_t t = (uint64_t)j * G + g;
            eval_lc_tiled((fe *)(a + 4 * t), inst->a_ptr, inst->a_col, inst->a_val, g, Z, inst->num_input, in_off, aux_off, num_input, NULL, a_aux);
            eval_lc_tiled((fe *)(b + 4 * t), inst->b_ptr, inst->b_col, inst->b_val, g, Z, inst->num_input, in_off, aux_off, num_input, b_input, b_aux);
            eval_lc_tiled((fe *)(c + 4 * t), inst->c_ptr, inst->c_col, inst->c_val, g, Z, inst->num_input, in_off, aux_off, num_input, NULL, NULL);
        }
    }
    for (uint32_t i = 0; i < num_input; i++) {   /* input_i * 0 = 0 */
        uint64_t row = (uint64_t)copies * G + i;
        memcpy(a + 4 * row, Z + i, 32); memset(b + 4 * row, 0, 32); memset(c + 4 * row, 0, 32);
    }
}

/* ------------------------------------------------------------------ key material */
typedef struct {
    uint64_t m; uint32_t num_input, num_aux;
    uint64_t n_h, n_l, n_a, n_b;            /* element counts of h, l, a, b_g1/b_g2 */
    uint8_t alpha_g1[64], beta_g1[64], beta_g2[128], gamma_g2[128], delta_g1[64], delta_g2[128];
    uint8_t *ic;    /* num_input x 64  */
    uint8_t *h;     /* (m-1) x 64      */
    uint8_t *l;     /* num_aux x 64    */
    uint8_t *a;     /* n_a x 64        */
    uint8_t *b_g1;  /* n_b x 64        */
    uint8_t *b_g2;  /* n_b x 128       */
} orc_key;

void orc_key_free(orc_key *k) { if (!k) return; free(k->ic); free(k->h); free(k->l); free(k->a); free(k->b_g1); free(k->b_g2); free(k); }

/* generate_parameters restated (Appendix A.4).  Toxic waste as Montgomery Fr. */
orc_key *orc_setup(const orc_r1cs *cs, const uint64_t tau_[4], const uint64_t alpha_[4], const uint64_t beta_[4],
                   const uint64_t gamma_[4], const uint64_t delta_[4], const uint8_t g1_[64], const uint8_t g2_[128]) {
    orc_init();
    const fe *tau = (const fe *)tau_, *alpha = (const fe *)alpha_, *beta = (const fe *)beta_, *gamma = (const fe *)gamma_, *delta = (const fe *)delta_;
    uint64_t n = cs->num_gates + cs->num_input;
    domain_t d; if (domain_init(&d, n)) return NULL;
    uint64_t m = d.m;
    g1_aff g1; g2_aff g2; g1_load(&g1, g1_); g2_load(&g2, g2_);
    g1_fbtable T1; g2_fbtable T2; g1_fb_init(&T1, &g1); g2_fb_init(&T2, &g2);
    fe gamma_inv, delta_inv; fe_inv(&FR, &gamma_inv, gamma); fe_inv(&FR, &delta_inv, delta);

    orc_key *K = (orc_key *)calloc(1, sizeof(orc_key));
    K->m = m; K->num_input = cs->num_input; K->num_aux = cs->num_aux;
    /* powers of tau */
    fe *pt = (fe *)malloc(m * sizeof(fe));
    { fe cur = FR.r; for (uint64_t i = 0; i < m; i++) { pt[i] = cur; fe_mul(&FR, &cur, &cur, tau); } }
    /* h[i] = g1 * (tau^i * (tau^m - 1)/delta), i < m-1 */
    fe coeff; { fe tm; fr_pow_u64(&tm, tau, m); fe_sub(&FR, &coeff, &tm, &FR.r); fe_mul(&FR, &coeff, &coeff, &delta_inv); }
    K->n_h = m - 1; K->h = (uint8_t *)malloc((m - 1) * 64 + 64);
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < m - 1; i++) {
        fe e; fe_mul(&FR, &e, &pt[i], &coeff); uint64_t k[4]; fe_to_canon(&FR, k, &e);
        g1_aff p; g1_fb_mul(&p, &T1, k); g1_store(K->h + 64 * i, &p);
    }
    /* Lagrange coefficients at tau */
    dom_ifft(&d, pt);
    uint32_t nv = cs->num_input + cs->num_aux;
    fe *At = (fe *)calloc(nv, sizeof(fe)), *Bt = (fe *)calloc(nv, sizeof(fe)), *Ct = (fe *)calloc(nv, sizeof(fe));
    for (uint64_t g = 0; g < cs->num_gates; g++) {
        for (uint64_t k = cs->a_ptr[g]; k < cs->a_ptr[g + 1]; k++) { fe t; fe_mul(&FR, &t, (const fe *)(cs->a_val + 4 * k), &pt[g]); fe_add(&FR, &At[cs->a_col[k]], &At[cs->a_col[k]], &t); }
        for (uint64_t k = cs->b_ptr[g]; k < cs->b_ptr[g + 1]; k++) { fe t; fe_mul(&FR, &t, (const fe *)(cs->b_val + 4 * k), &pt[g]); fe_add(&FR, &Bt[cs->b_col[k]], &Bt[cs->b_col[k]], &t); }
        for (uint64_t k = cs->c_ptr[g]; k < cs->c_ptr[g + 1]; k++) { fe t; fe_mul(&FR, &t, (const fe *)(cs->c_val + 4 * k), &pt[g]); fe_add(&FR, &Ct[cs->c_col[k]], &Ct[cs->c_col[k]], &t); }
    }
    for (uint32_t i = 0; i < cs->num_input; i++) fe_add(&FR, &At[i], &At[i], &pt[cs->num_gates + i]);

    uint8_t *a_all = (uint8_t *)malloc((size_t)nv * 64), *b1_all = (uint8_t *)malloc((size_t)nv * 64), *b2_all = (uint8_t *)malloc((size_t)nv * 128);
    K->ic = (uint8_t *)malloc((size_t)cs->num_input * 64 + 64);
    K->l = (uint8_t *)malloc((size_t)cs->num_aux * 64 + 64);
    K->n_l = cs->num_aux;
    #pragma omp parallel for schedule(static)
    for (uint32_t v = 0; v < nv; v++) {
        uint64_t k[4]; g1_aff p; g2_aff p2;
        fe_to_canon(&FR, k, &At[v]); g1_fb_mul(&p, &T1, k); g1_store(a_all + 64 * (size_t)v, &p);
        fe_to_canon(&FR, k, &Bt[v]); g1_fb_mul(&p, &T1, k); g1_store(b1_all + 64 * (size_t)v, &p);
        g2_fb_mul(&p2, &T2, k); g2_store(b2_all + 128 * (size_t)v, &p2);
        fe e, t; fe_mul(&FR, &e, &At[v], beta); fe_mul(&FR, &t, &Bt[v], alpha); fe_add(&FR, &e, &e, &t); fe_add(&FR, &e, &e, &Ct[v]);
        fe_mul(&FR, &e, &e, v < cs->num_input ? &gamma_inv : &delta_inv);
        fe_to_canon(&FR, k, &e); g1_fb_mul(&p, &T1, k);
        if (v < cs->num_input) g1_store(K->ic + 64 * (size_t)v, &p); else g1_store(K->l + 64 * (size_t)(v - cs->num_input), &p);
    }
    /* a, b_g1, b_g2 filtered to non-identity points (inputs first, then aux) */
    static const uint8_t zero128[128] = {0};
    K->a = (uint8_t *)malloc((size_t)nv * 64 + 64); K->b_g1 = (uint8_t *)malloc((size_t)nv * 64 + 64); K->b_g2 = (uint8_t *)malloc((size_t)nv * 128 + 128);
    uint64_t nb2 = 0;
    for (uint32_t v = 0; v < nv; v++) {
        if (memcmp(a_all + 64 * (size_t)v, zero128, 64)) { memcpy(K->a + 64 * K->n_a, a_all + 64 * (size_t)v, 64); K->n_a++; }
        if (memcmp(b1_all + 64 * (size_t)v, zero128, 64)) { memcpy(K->b_g1 + 64 * K->n_b, b1_all + 64 * (size_t)v, 64); K->n_b++; }
        if (memcmp(b2_all + 128 * (size_t)v, zero128, 128)) { memcpy(K->b_g2 + 128 * nb2, b2_all + 128 * (size_t)v, 128); nb2++; }
    }
    if (nb2 != K->n_b) { fprintf(stderr, "orc_setup: b_g1/b_g2 count mismatch\n"); }
    /* vk */
    { uint64_t k[4]; g1_aff p; g2_aff p2;
      fe_to_canon(&FR, k, alpha); g1_fb_mul(&p, &T1, k); g1_store(K->alpha_g1, &p);
      fe_to_canon(&FR, k, beta); g1_fb_mul(&p, &T1, k); g1_store(K->beta_g1, &p); g2_fb_mul(&p2, &T2, k); g2_store(K->beta_g2, &p2);
      fe_to_canon(&FR, k, gamma); g2_fb_mul(&p2, &T2, k); g2_store(K->gamma_g2, &p2);
      fe_to_canon(&FR, k, delta); g1_fb_mul(&p, &T1, k); g1_store(K->delta_g1, &p); g2_fb_mul(&p2, &T2, k); g2_store(K->delta_g2, &p2); }
    free(pt); free(At); free(Bt); free(Ct); free(a_all); free(b1_all); free(b2_all); free(T1.t); free(T2.t);
    return K;
}

/* create_proof(circuit, params, r, s) restated (Appendix A.1-A.5).  a,b,c,densities come from
 * orc_synthesize; z = z_input || z_aux; r,s Montgomery Fr.  out: 256-byte fawkes Borsh proof
 * (prover.rs:39-45; canonical LE coordinates; all-zero = infinity).  Also returns the five MSM
 * results (raw LE) in `msm_out` (4 x 64 + 128 B: H, L, A, B1, B2) when non-NULL.
 * Returns 0, or -1 PolynomialDegreeTooLarge, -2 UnexpectedIdentity, -3 key/density size mismatch. */
int orc_prove(const orc_key *K, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n,
              const uint64_t *z, const uint8_t *a_aux, const uint8_t *b_input, const uint8_t *b_aux,
              const uint64_t r_[4], const uint64_t s_[4], uint8_t out[256], uint8_t *msm_out) {
    orc_init();
    uint32_t v_in = K->num_input, v_aux = K->num_aux;
    domain_t d; if (domain_init(&d, n)) return -1;
    if (d.m != K->m) return -3;
    uint64_t n_a_aux = 0, n_b_in = 0, n_b_aux = 0;
    for (uint32_t j = 0; j < v_aux; j++) { n_a_aux += a_aux[j] ? 1 : 0; n_b_aux += b_aux[j] ? 1 : 0; }
    for (uint32_t i = 0; i < v_in; i++) n_b_in += b_input[i] ? 1 : 0;
    if (v_in + n_a_aux != K->n_a || n_b_in + n_b_aux != K->n_b) return -3;

    uint64_t *h = (uint64_t *)malloc(d.m * 32);
    if (!orc_quotient_h(a, b, c, n, h)) { free(h); return -1; }
    uint64_t *hc = scalars_to_canon(h, d.m - 1);
    uint64_t *zc = scalars_to_canon(z, (size_t)v_in + v_aux);
    const uint64_t *zin = zc, *zaux = zc + 4 * (size_t)v_in;

    /* the eight multiexps of bellman's prover (H; L; A over inputs / aux; B1 and B2 over inputs / aux).  Each is a list of
     * independent regions; bellman queues them all on its worker pool at once (create_proof starts every multiexp before it
     * waits for any), so they run here as ONE task list -- serially in order with one thread. */
    g1_aff *bh = (g1_aff *)malloc((K->n_h + 1) * sizeof(g1_aff)), *bl = (g1_aff *)malloc((K->n_l + 1) * sizeof(g1_aff));
    g1_aff *ba = (g1_aff *)malloc((K->n_a + 1) * sizeof(g1_aff)), *bb1 = (g1_aff *)malloc((K->n_b + 1) * sizeof(g1_aff));
    g2_aff *bb2 = (g2_aff *)malloc((K->n_b + 1) * sizeof(g2_aff));
    #pragma omp parallel num_threads(ORC_THREADS)
    {
        #pragma omp for schedule(static) nowait
        for (uint64_t i = 0; i < K->n_h; i++) g1_load(&bh[i], K->h + 64 * i);
        #pragma omp for schedule(static) nowait
        for (uint64_t i = 0; i < K->n_l; i++) g1_load(&bl[i], K->l + 64 * i);
        #pragma omp for schedule(static) nowait
        for (uint64_t i = 0; i < K->n_a; i++) g1_load(&ba[i], K->a + 64 * i);
        #pragma omp for schedule(static) nowait
        for (uint64_t i = 0; i < K->n_b; i++) g1_load(&bb1[i], K->b_g1 + 64 * i);
        #pragma omp for schedule(static)
        for (uint64_t i = 0; i < K->n_b; i++) g2_load(&bb2[i], K->b_g2 + 128 * i);
    }
    struct mexp { int g2; const void *bases; const uint8_t *dens; const uint64_t *exps; size_t n; unsigned c, nreg, first; } me[8] = {
        {0, bh, NULL, hc, d.m - 1, 0, 0, 0}, {0, bl, NULL, zaux, v_aux, 0, 0, 0},
        {0, ba, NULL, zin, v_in, 0, 0, 0}, {0, ba + v_in, a_aux, zaux, v_aux, 0, 0, 0},
        {0, bb1, b_input, zin, v_in, 0, 0, 0}, {0, bb1 + n_b_in, b_aux, zaux, v_aux, 0, 0, 0},
        {1, bb2, b_input, zin, v_in, 0, 0, 0}, {1, bb2 + n_b_in, b_aux, zaux, v_aux, 0, 0, 0}};
    unsigned ntask = 0;
    for (int k = 0; k < 8; k++) { me[k].c = g1_multiexp_window(me[k].n); me[k].nreg = g1_multiexp_regions(me[k].c); me[k].first = ntask; ntask += me[k].nreg; }
    g1_jac *reg1 = (g1_jac *)malloc(ntask * sizeof(g1_jac)); g2_jac *reg2 = (g2_jac *)malloc(ntask * sizeof(g2_jac));
    #pragma omp parallel for schedule(dynamic, 1) num_threads(ORC_THREADS)
    for (unsigned t = 0; t < ntask; t++) {
        int k = 7; while (me[k].first > t) k--;
        const unsigned reg = t - me[k].first;
        if (me[k].g2) g2_multiexp_region(&reg2[t], (const g2_aff *)me[k].bases, me[k].dens, me[k].exps, me[k].n, me[k].c, reg);
        else g1_multiexp_region(&reg1[t], (const g1_aff *)me[k].bases, me[k].dens, me[k].exps, me[k].n, me[k].c, reg);
    }
    g1_jac H, L, Ain, Aaux, B1in, B1aux; g2_jac B2in, B2aux;
    g1_jac *o1[6] = {&H, &L, &Ain, &Aaux, &B1in, &B1aux}; g2_jac *o2[2] = {&B2in, &B2aux};
    for (int k = 0; k < 6; k++) g1_multiexp_join(o1[k], reg1 + me[k].first, me[k].nreg, me[k].c);
    for (int k = 6; k < 8; k++) g2_multiexp_join(o2[k - 6], reg2 + me[k].first, me[k].nreg, me[k].c);
    free(reg1); free(reg2);

    g1_aff alpha1, beta1, delta1; g2_aff beta2, delta2;
    g1_load(&alpha1, K->alpha_g1); g1_load(&beta1, K->beta_g1); g1_load(&delta1, K->delta_g1);
    g2_load(&beta2, K->beta_g2); g2_load(&delta2, K->delta_g2);
    int rc = 0;
    if (delta1.inf || delta2.inf) rc = -2;
    else {
        uint64_t r[4], s[4], rs[4]; fe rsm;
        fe_to_canon(&FR, r, (const fe *)r_); fe_to_canon(&FR, s, (const fe *)s_);
        fe_mul(&FR, &rsm, (const fe *)r_, (const fe *)s_); fe_to_canon(&FR, rs, &rsm);
        g1_jac d1, a1, b1j, t; g2_jac d2;
        g1_jac_from_aff(&d1, &delta1); g1_jac_from_aff(&a1, &alpha1); g1_jac_from_aff(&b1j, &beta1); g2_jac_from_aff(&d2, &delta2);
        g1_jac g_a, g_c; g2_jac g_b;
        g1_jac_mul(&g_a, &d1, r); g1_jac_add_mixed(&g_a, &alpha1);
        g2_jac_mul(&g_b, &d2, s); g2_jac_add_mixed(&g_b, &beta2);
        g1_jac_mul(&g_c, &d1, rs);
        g1_jac_mul(&t, &a1, s); g1_jac_add(&g_c, &t);
        g1_jac_mul(&t, &b1j, r); g1_jac_add(&g_c, &t);
        g1_jac a_ans = Ain; g1_jac_add(&a_ans, &Aaux);
        g1_jac_add(&g_a, &a_ans);
        g1_jac_mul(&t, &a_ans, s); g1_jac_add(&g_c, &t);
        g1_jac b1_ans = B1in; g1_jac_add(&b1_ans, &B1aux);
        g2_jac b2_ans = B2in; g2_jac_add(&b2_ans, &B2aux);
        g2_jac_add(&g_b, &b2_ans);
        g1_jac_mul(&t, &b1_ans, r); g1_jac_add(&g_c, &t);
        g1_jac_add(&g_c, &H); g1_jac_add(&g_c, &L);
        g1_aff pa, pc; g2_aff pb;
        g1_jac_to_aff(&pa, &g_a); g2_jac_to_aff(&pb, &g_b); g1_jac_to_aff(&pc, &g_c);
        memset(out, 0, 256);
        if (!pa.inf) { fe_to_canon(&FQ, (uint64_t *)(out), &pa.x); fe_to_canon(&FQ, (uint64_t *)(out + 32), &pa.y); }
        if (!pb.inf) {
            fe_to_canon(&FQ, (uint64_t *)(out + 64), &pb.x.c0); fe_to_canon(&FQ, (uint64_t *)(out + 96), &pb.x.c1);
            fe_to_canon(&FQ, (uint64_t *)(out + 128), &pb.y.c0); fe_to_canon(&FQ, (uint64_t *)(out + 160), &pb.y.c1);
        }
        if (!pc.inf) { fe_to_canon(&FQ, (uint64_t *)(out + 192), &pc.x); fe_to_canon(&FQ, (uint64_t *)(out + 224), &pc.y); }
        if (msm_out) {
            g1_aff t1; g2_aff t2;
            g1_jac_to_aff(&t1, &H); g1_store(msm_out, &t1);
            g1_jac_to_aff(&t1, &L); g1_store(msm_out + 64, &t1);
            g1_jac_to_aff(&t1, &a_ans); g1_store(msm_out + 128, &t1);
            g1_jac_to_aff(&t1, &b1_ans); g1_store(msm_out + 192, &t1);
            g2_jac_to_aff(&t2, &b2_ans); g2_store(msm_out + 256, &t2);
        }
    }
    free(h); free(hc); free(zc); free(bh); free(bl); free(ba); free(bb1); free(bb2);
    return rc;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
