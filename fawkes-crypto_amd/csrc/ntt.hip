// BN254 Fr radix-2 NTT / iNTT over bellman's evaluation domain and the A*B-C quotient pipeline.
//
// Replaces bellman_ce::domain::EvaluationDomain::{ifft, coset_fft, mul_assign, sub_assign,
// divide_by_z_on_coset, icoset_fft} as driven by bellman's prover (reached from
// /root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80).  Convention: natural order
// in and out, out[k] = sum_j in[j] * omega^(jk), omega = ROOT_OF_UNITY^(2^(S-exp)), generator 7
// (fawkes-crypto/src/engines/bn256/mod.rs:23-24).
//
// MI355X design: a Stockham autosort decomposition (no bit-reversal pass over HBM).  The transform of
// size N = 2^k is cut into P = ceil(k/9) passes; pass i does radix-2^deg_i sub-transforms entirely in
// LDS.  A workgroup owns a tile of R = 2^deg rows x C adjacent columns so that every HBM access is a
// run of C*32 contiguous bytes (coalesced 16 B/lane loads), stages it in LDS as two 16-byte planes
// (conflict-free ds_read_b128/ds_write_b128), runs deg butterfly stages (decimation in frequency,
// bit-reversed inside the tile only), and writes the tile to its autosorted position.  Inter-pass
// twiddles omega^(k*r*N/(pR)) come from a two-level table (2^L + 2^(k-L) entries, L2-resident).
// The coset shift (g^i), the 1/m scaling, the 1/Z(g) factor and the pointwise a*b-c are fused into
// the first/last pass of the neighbouring transform; the whole quotient is 6 transforms (quotient_dev: c is subtracted in
// coefficient space, bellman does 7) = 6*P passes, each reading and writing every element once (64 B/element/pass).
#include "common.hpp"

namespace fk {

static constexpr uint32_t NTT_MAXDEG = 9;       // R <= 512
#ifndef FK_NTT_TILE_LOG
#define FK_NTT_TILE_LOG 10
#endif
static constexpr uint32_t NTT_TILE_LOG = FK_NTT_TILE_LOG;    // R*C <= 1024 elements = 32 KiB of LDS: four workgroups per compute unit (2048 elements / two: 169.4 -> 168.1 ms per proof; 4096 / one: 171.2; profiles/r02_sorts_first_probe.log)
static constexpr uint32_t NTT_MAX_THREADS = 1024;

enum { PRE_NONE = 0, PRE_TABLE = 1, PRE_ABC = 2, PRE_AB = 3 };
enum { POST_NONE = 0, POST_CONST = 1, POST_TABLE = 2, POST_TABLE_SUB = 3 };      // _SUB: v * table - xc[out index]

struct ScaleTable {          // value(i) = lo[i & (2^L - 1)] * hi[i >> L]
    Fr *lo = nullptr, *hi = nullptr;
    Fr *full = nullptr;      // the products themselves (the quotient's two post-scale tables, memory permitting): one product per element instead of two
};

struct NttDomain {
    uint32_t log_n = 0, L = 0, maxdeg = 0;
    std::vector<uint32_t> degs;
    Fr omega, omega_inv, minv, zinv;
    Fr *tw_lo[2] = {nullptr, nullptr}, *tw_hi[2] = {nullptr, nullptr};   // [0] forward, [1] inverse
    Fr *pq[2] = {nullptr, nullptr};                                       // omega_Rmax^e, e < Rmax/2
    ScaleTable t_g, t_ginv_minv, t_g_minv, t_ginv_minv_zinv;
    // single-level inter-pass twiddles: tw_full[dir][pass][k * r] = w^((k * r) << (log_n - lgp - deg)); one product per
    // element instead of two (lo * hi, then the element).  nullptr: fall back to the two-level tables.
    std::vector<Fr *> tw_full[2];
    std::map<uint64_t, ScaleTable> dist_post;   // distributed quotient: (omega_m^-rank)^k tables, key = log_w << 32 | rank
    std::vector<void *> allocs;
};

struct PassArgs {
    const Fr *x, *xb, *xc;
    Fr *y;
    uint32_t log_n, deg, lgp, logC;
    const Fr *tw_lo, *tw_hi;
    const Fr *tw_full;
    uint32_t L;
    const Fr *pq;
    uint32_t pq_shift;
    int pre_mode;
    const Fr *pre_lo, *pre_hi;
    int post_mode;
    Fr post_const;
    const Fr *post_lo, *post_hi, *post_full;
};

// The passes compute in the lazily reduced form FrL ([0, 2r): no conditional subtraction behind a product, field.hpp); what
// they read from memory (data, tables) is canonical, hence valid, and what they store is made canonical again.
// (FL = Fr itself: the canonical form, FK_NTT_LAZY=0.)
template <class FL> static __device__ __forceinline__ FL ldl_(const Fr &x) { FL r; for (int i = 0; i < 8; i++) r.v[i] = x.v[i]; return r; }
static __device__ __forceinline__ Fr canon(const Fr &x) { return x; }
template <class FL>
static __device__ __forceinline__ void lds_put(uint4 *p0, uint4 *p1, uint32_t e, const FL &v) {
    p0[e] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
    p1[e] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
}
template <class FL>
static __device__ __forceinline__ FL lds_get(const uint4 *p0, const uint4 *p1, uint32_t e) {
    uint4 a = p0[e], b = p1[e];
    FL v; v.v[0] = a.x; v.v[1] = a.y; v.v[2] = a.z; v.v[3] = a.w; v.v[4] = b.x; v.v[5] = b.y; v.v[6] = b.z; v.v[7] = b.w;
    return v;
}

template <class FL>
__global__ __launch_bounds__(NTT_MAX_THREADS) void ntt_pass_kernel(PassArgs a) {
    auto ldl = [](const Fr *p, uint64_t i) { return ldl_<FL>(p[i]); };
    const uint32_t NTT_THREADS = blockDim.x;
    extern __shared__ uint4 lds[];
    const uint32_t R = 1u << a.deg, C = 1u << a.logC, tile = R << a.logC;
    uint4 *p0 = lds, *p1 = lds + tile;
    const uint32_t tid = threadIdx.x;
    const uint64_t t = (uint64_t)1 << (a.log_n - a.deg);        // stride between the R inputs
    const uint64_t pmask = ((uint64_t)1 << a.lgp) - 1;
    const uint64_t i_base = (uint64_t)blockIdx.x << a.logC;
    const uint32_t Lmask = (1u << a.L) - 1;

    // Every lane handles two elements / butterflies per step (e and e + NTT_THREADS) so that the field products can
    // be issued as dual chains (FL::mul2: two interleaved accumulator chains per wave, the same instruction count).
    // tile is a multiple of 2 * NTT_THREADS except for tiny transforms, where the second element is masked off.

    // ---- load tile (rows r, columns c), fused pre-op and inter-pass twiddle
    
    for (uint32_t e0 = tid; e0 < tile; e0 += 2 * NTT_THREADS) {
        const uint32_t e1 = e0 + NTT_THREADS;
        const bool has1 = e1 < tile;
        const uint32_t ee1 = has1 ? e1 : e0;
        const uint32_t c0 = e0 & (C - 1), r0 = e0 >> a.logC, c1 = ee1 & (C - 1), r1 = ee1 >> a.logC;
        const uint64_t i0 = i_base + c0, i1 = i_base + c1;
        const uint64_t idx0 = i0 + (uint64_t)r0 * t, idx1 = i1 + (uint64_t)r1 * t;
        FL v0 = ldl(a.x, idx0), v1 = ldl(a.x, idx1);
        if (a.pre_mode == PRE_ABC) {
            FL p0, p1;
            FL::mul2(v0, ldl(a.xb, idx0), v1, ldl(a.xb, idx1), p0, p1);
            FL::sub2(p0, ldl(a.xc, idx0), p1, ldl(a.xc, idx1), v0, v1);
        } else if (a.pre_mode == PRE_AB) {
            FL::mul2(v0, ldl(a.xb, idx0), v1, ldl(a.xb, idx1), v0, v1);
        } else if (a.pre_mode == PRE_TABLE) {
            FL s0, s1;
            FL::mul2(ldl(a.pre_lo, idx0 & Lmask), ldl(a.pre_hi, idx0 >> a.L), ldl(a.pre_lo, idx1 & Lmask), ldl(a.pre_hi, idx1 >> a.L), s0, s1);
            FL::mul2(v0, s0, v1, s1, v0, v1);
        }
        if (a.lgp) {
            const uint64_t y0 = (i0 & pmask) * r0, y1 = (i1 & pmask) * r1;
            if (a.tw_full) {
                FL::mul2(v0, ldl(a.tw_full, y0), v1, ldl(a.tw_full, y1), v0, v1);
            } else {
                const uint64_t x0 = y0 << (a.log_n - a.lgp - a.deg), x1 = y1 << (a.log_n - a.lgp - a.deg);
                FL w0, w1;
                FL::mul2(ldl(a.tw_lo, x0 & Lmask), ldl(a.tw_hi, x0 >> a.L), ldl(a.tw_lo, x1 & Lmask), ldl(a.tw_hi, x1 >> a.L), w0, w1);
                FL::mul2(v0, w0, v1, w1, v0, v1);
            }
        }
        lds_put(p0, p1, e0, v0);
        if (has1) lds_put(p0, p1, e1, v1);
    }
    __syncthreads();

    // ---- deg DIF stages in LDS, two at a time: a lane holds the four elements of a radix-4 group in registers, so a
    // pair of stages costs one LDS round trip and one barrier (same four products as two radix-2 stages:
    // T1, T2 for the first stage, T3 twice for the second -- issued as two dual chains)
    const uint32_t ngrp = tile >> 2;
    uint32_t rnd = 0;
    for (; rnd + 1 < a.deg; rnd += 2) {
        const uint32_t bit = (R >> 1) >> rnd, bq = bit >> 1;
        for (uint32_t g = tid; g < ngrp; g += NTT_THREADS) {
            const uint32_t c = g & (C - 1), ii = g >> a.logC;
            const uint32_t d = ii & (bq - 1);
            const uint32_t row = ((ii - d) << 2) + d;                 // bits `bit` and `bq` of the row are zero
            const uint32_t e0 = (row << a.logC) + c, e1 = e0 + (bq << a.logC), e2 = e0 + (bit << a.logC), e3 = e2 + (bq << a.logC);
            FL x0 = lds_get<FL>(p0, p1, e0), x1 = lds_get<FL>(p0, p1, e1), x2 = lds_get<FL>(p0, p1, e2), x3 = lds_get<FL>(p0, p1, e3);
            FL s0, d0, s1, d1, y0, y1;
            FL::addsub2(x0, x2, x0, x2, s0, d0);
            FL::addsub2(x1, x3, x1, x3, s1, d1);
            FL::addsub2(s0, s1, s0, s1, y0, y1);
            if (bq == 1) {                                            // d = 0: T1 = T3 = 1, T2 = the fourth root
                d1 = FL::mul(d1, ldl(a.pq, (1u << rnd) << a.pq_shift));
            } else {
                FL::mul2(d0, ldl(a.pq, (d << rnd) << a.pq_shift), d1, ldl(a.pq, ((d + bq) << rnd) << a.pq_shift), d0, d1);
            }
            FL y2, y3;
            FL::addsub2(d0, d1, d0, d1, y2, y3);
            if (bq != 1) {
                const FL t3 = ldl(a.pq, (d << (rnd + 1)) << a.pq_shift);
                FL::mul2(y1, t3, y3, t3, y1, y3);
            }
            lds_put(p0, p1, e0, y0);
            lds_put(p0, p1, e1, y1);
            lds_put(p0, p1, e2, y2);
            lds_put(p0, p1, e3, y3);
        }
        __syncthreads();
    }
    if (rnd < a.deg) {                        // odd degree: the last stage stands alone, and all its twiddles are 1
        const uint32_t nbf = tile >> 1;
        for (uint32_t bf = tid; bf < nbf; bf += NTT_THREADS) {
            const uint32_t c = bf & (C - 1), ii = bf >> a.logC;
            const uint32_t e0 = ((ii << 1) << a.logC) + c, e1 = e0 + C;
            FL u = lds_get<FL>(p0, p1, e0), w = lds_get<FL>(p0, p1, e1), sm, df;
            FL::addsub2(u, w, u, w, sm, df);
            lds_put(p0, p1, e0, sm);
            lds_put(p0, p1, e1, df);
        }
        __syncthreads();
    }

    // ---- store tile to its autosorted place, fused post-op
    for (uint32_t e0 = tid; e0 < tile; e0 += 2 * NTT_THREADS) {
        const uint32_t e1_ = e0 + NTT_THREADS;
        const bool has1 = e1_ < tile;
        const uint32_t e1 = has1 ? e1_ : e0;
        uint32_t c0, rr0, c1, rr1;
        if (a.lgp == 0) { rr0 = e0 & (R - 1); c0 = e0 >> a.deg; rr1 = e1 & (R - 1); c1 = e1 >> a.deg; }      // first pass: runs of R outputs
        else { c0 = e0 & (C - 1); rr0 = e0 >> a.logC; c1 = e1 & (C - 1); rr1 = e1 >> a.logC; }            // later passes: runs of C outputs
        const uint32_t q0 = a.deg ? (__brev(rr0) >> (32 - a.deg)) : 0, q1 = a.deg ? (__brev(rr1) >> (32 - a.deg)) : 0;
        FL v0 = lds_get<FL>(p0, p1, (q0 << a.logC) + c0), v1 = lds_get<FL>(p0, p1, (q1 << a.logC) + c1);
        const uint64_t i0 = i_base + c0, i1 = i_base + c1;
        const uint64_t k0 = i0 & pmask, k1 = i1 & pmask;
        const uint64_t o0 = ((i0 - k0) << a.deg) + k0 + ((uint64_t)rr0 << a.lgp), o1 = ((i1 - k1) << a.deg) + k1 + ((uint64_t)rr1 << a.lgp);
        if (a.post_mode == POST_CONST) { const FL pc = ldl_<FL>(a.post_const); FL::mul2(v0, pc, v1, pc, v0, v1); }
        else if (a.post_mode == POST_TABLE || a.post_mode == POST_TABLE_SUB) {
            if (a.post_full) FL::mul2(v0, ldl(a.post_full, o0), v1, ldl(a.post_full, o1), v0, v1);
            else {
                FL s0, s1;
                FL::mul2(ldl(a.post_lo, o0 & Lmask), ldl(a.post_hi, o0 >> a.L), ldl(a.post_lo, o1 & Lmask), ldl(a.post_hi, o1 >> a.L), s0, s1);
                FL::mul2(v0, s0, v1, s1, v0, v1);
            }
            if (a.post_mode == POST_TABLE_SUB) FL::sub2(v0, ldl(a.xc, o0), v1, ldl(a.xc, o1), v0, v1);
        }
        a.y[o0] = canon(v0);
        if (has1) a.y[o1] = canon(v1);
    }
}

__global__ void fr_mul_batch_kernel(const Fr *a, const Fr *b, Fr *o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = Fr::mul(a[i], b[i]);
}

__global__ void tw_full_kernel(const Fr *lo, const Fr *hi, uint32_t L, uint32_t shift, uint64_t n, Fr *out) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint64_t e = x << shift;
    out[x] = Fr::mul(lo[e & ((1u << L) - 1)], hi[e >> L]);
}

int fr_mul_batch_dev(fk_ctx *ctx, const Fr *a, const Fr *b, Fr *o, size_t n) {
    if (!n) return FK_OK;
    hipLaunchKernelGGL(fr_mul_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, b, o, n);
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}

// ------------------------------------------------------------------------------------------ domain setup (host)
static Fr host_const(const uint32_t (&w)[8]) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = w[i]; return r; }

static int upload_table(fk_ctx *ctx, NttDomain *d, const std::vector<Fr> &h, Fr **out) {
    void *p = nullptr;
    FK_HIP(ctx, hipMalloc(&p, h.size() * sizeof(Fr)));
    d->allocs.push_back(p);
    FK_HIP(ctx, hipMemcpy(p, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice));
    *out = (Fr *)p;
    return FK_OK;
}

// lo[i] = c * base^i (i < 2^L), hi[j] = base^(j 2^L) (j < 2^(k-L))
static int make_scale_table(fk_ctx *ctx, NttDomain *d, const Fr &base, const Fr &c, ScaleTable *t) {
    const uint32_t nlo = 1u << d->L, nhi = 1u << (d->log_n - d->L);
    std::vector<Fr> lo(nlo), hi(nhi);
    Fr cur = c, pw = Fr::one();
    for (uint32_t i = 0; i < nlo; i++) { lo[i] = cur; cur = Fr::mul(cur, base); pw = Fr::mul(pw, base); }
    // pw = base^(2^L)
    cur = Fr::one();
    for (uint32_t j = 0; j < nhi; j++) { hi[j] = cur; cur = Fr::mul(cur, pw); }
    FK_TRY(upload_table(ctx, d, lo, &t->lo));
    FK_TRY(upload_table(ctx, d, hi, &t->hi));
    return FK_OK;
}

static int get_domain(fk_ctx *ctx, uint32_t log_n, NttDomain **out) {
    auto it = ctx->domains.find(log_n);
    if (it != ctx->domains.end()) { *out = it->second; return FK_OK; }
    // bellman's EvaluationDomain::from_coeffs rejects exp >= S
    if (log_n >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_n, FK_FR_S - 1);
    NttDomain *d = new NttDomain();
    d->log_n = log_n;
    d->L = (log_n + 1) / 2;
    const uint32_t P = log_n ? (log_n + NTT_MAXDEG - 1) / NTT_MAXDEG : 1;
    const uint32_t base = log_n / P, extra = log_n % P;
    for (uint32_t i = 0; i < P; i++) d->degs.push_back(base + (i < extra ? 1 : 0));
    d->maxdeg = d->degs[0];
    const uint32_t root_w[8] = FK_FR_ROOT, gen_w[8] = FK_FR_GEN, geninv_w[8] = FK_FR_GEN_INV;
    Fr om = host_const(root_w);
    for (uint32_t i = log_n; i < FK_FR_S; i++) om = Fr::sqr(om);
    d->omega = om;
    d->omega_inv = Fr::inv(om);
    d->minv = Fr::inv(Fr::from_u64((uint64_t)1 << log_n));
    const Fr g = host_const(gen_w), ginv = host_const(geninv_w);
    d->zinv = Fr::inv(Fr::sub(Fr::pow_u64(g, (uint64_t)1 << log_n), Fr::one()));
    int rc = FK_OK;
    for (int dir = 0; dir < 2 && rc == FK_OK; dir++) {
        const Fr w = dir ? d->omega_inv : d->omega;
        ScaleTable t;
        rc = make_scale_table(ctx, d, w, Fr::one(), &t);
        d->tw_lo[dir] = t.lo; d->tw_hi[dir] = t.hi;
        if (rc != FK_OK) break;
        // pq[e] = (w^(N/Rmax))^e
        const uint32_t half = d->maxdeg ? (1u << (d->maxdeg - 1)) : 1;
        Fr wr = w;
        for (uint32_t i = d->maxdeg; i < log_n; i++) wr = Fr::sqr(wr);
        std::vector<Fr> pq(half);
        Fr cur = Fr::one();
        for (uint32_t e = 0; e < half; e++) { pq[e] = cur; cur = Fr::mul(cur, wr); }
        rc = upload_table(ctx, d, pq, &d->pq[dir]);
    }
    // single-level twiddles for every pass but the first (sizes 2^(lgp + deg): the last one is the whole domain, 32 B per
    // point and direction).  FK_NTT_FULL_TW=0 disables them; they are skipped when they would not fit comfortably.
    {
        const bool want = tune("FK_NTT_FULL_TW", 1) != 0;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        uint64_t need = 0;
        { uint32_t lgp = 0; for (size_t i = 0; i < d->degs.size(); i++) { if (i) need += (uint64_t)2 * sizeof(Fr) << (lgp + d->degs[i]); lgp += d->degs[i]; } }
        const bool fits = need < free_b / 8;
        for (int dir = 0; dir < 2 && rc == FK_OK; dir++) {
            d->tw_full[dir].assign(d->degs.size(), nullptr);
            if (!want || !fits) continue;
            uint32_t lgp = 0;
            for (size_t i = 0; i < d->degs.size() && rc == FK_OK; i++) {
                if (i) {
                    const uint64_t cnt = (uint64_t)1 << (lgp + d->degs[i]);
                    void *p = nullptr;
                    if (hipMalloc(&p, cnt * sizeof(Fr)) != hipSuccess) { rc = FK_ERR_OOM; break; }
                    d->allocs.push_back(p);
                    hipLaunchKernelGGL(tw_full_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, d->tw_lo[dir], d->tw_hi[dir], d->L,
                                       log_n - lgp - d->degs[i], cnt, (Fr *)p);
                    if (hipGetLastError() != hipSuccess) { rc = FK_ERR_HIP; break; }
                    d->tw_full[dir][i] = (Fr *)p;
                }
                lgp += d->degs[i];
            }
        }
        if (rc == FK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = FK_ERR_HIP;
    }
    if (rc == FK_OK) rc = make_scale_table(ctx, d, g, Fr::one(), &d->t_g);
    if (rc == FK_OK) rc = make_scale_table(ctx, d, ginv, d->minv, &d->t_ginv_minv);
    if (rc == FK_OK) rc = make_scale_table(ctx, d, g, d->minv, &d->t_g_minv);
    if (rc == FK_OK) rc = make_scale_table(ctx, d, ginv, Fr::mul(d->minv, d->zinv), &d->t_ginv_minv_zinv);
    // the quotient's two post-scale tables also in single-level form (2 x 32 B per point): their passes then spend one product per
    // element on the scale instead of two (3 of the ~83 products per point of a quotient)
    if (rc == FK_OK && tune("FK_NTT_FULL_SCALE", 1)) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const uint64_t cnt = (uint64_t)1 << log_n;
        if (2 * cnt * sizeof(Fr) < free_b / 8) {
            for (ScaleTable *t : {&d->t_g_minv, &d->t_ginv_minv_zinv}) {
                void *p = nullptr;
                if (hipMalloc(&p, cnt * sizeof(Fr)) != hipSuccess) { rc = FK_ERR_OOM; break; }
                d->allocs.push_back(p);
                hipLaunchKernelGGL(tw_full_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, t->lo, t->hi, d->L, 0u, cnt, (Fr *)p);
                if (hipGetLastError() != hipSuccess) { rc = FK_ERR_HIP; break; }
                t->full = (Fr *)p;
            }
            if (rc == FK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = FK_ERR_HIP;
        }
    }
    if (rc != FK_OK) { for (void *p : d->allocs) (void)hipFree(p); delete d; return rc; }
    ctx->domains[log_n] = d;
    *out = d;
    return FK_OK;
}

void ntt_free_domains(fk_ctx *ctx) {
    for (auto &kv : ctx->domains) { for (void *p : kv.second->allocs) (void)hipFree(p); delete kv.second; }
    ctx->domains.clear();
}

struct NttOp {
    bool inverse = false;
    int pre_mode = PRE_NONE; const ScaleTable *pre = nullptr; const Fr *xb = nullptr, *xc = nullptr;
    int post_mode = POST_NONE; Fr post_const; const ScaleTable *post = nullptr;
};

// Runs all passes: in -> ... -> final_dst, intermediates alternate between tmp1 and tmp2.
// Requirements: tmp1 != tmp2, neither aliases in / xb / xc / final_dst; final_dst may alias `in` only
// when there are >= 2 passes.
static int ntt_exec(fk_ctx *ctx, NttDomain *d, const NttOp &op, const Fr *in, Fr *tmp1, Fr *tmp2, Fr *final_dst) {
    const uint32_t P = (uint32_t)d->degs.size();
    const int dir = op.inverse ? 1 : 0;
    const Fr *src = in;
    uint32_t lgp = 0;
    for (uint32_t i = 0; i < P; i++) {
        const uint32_t deg = d->degs[i];
        Fr *dst;
        if (i == P - 1) dst = final_dst;
        else dst = (((P - 2 - i) & 1) == 0) ? tmp1 : tmp2;
        PassArgs a{};
        a.x = src; a.xb = op.xb; a.xc = op.xc; a.y = dst;
        a.log_n = d->log_n; a.deg = deg; a.lgp = lgp;
        uint32_t logC = 3;
        if (logC > NTT_TILE_LOG - deg) logC = NTT_TILE_LOG - deg;
        if (logC > d->log_n - deg) logC = d->log_n - deg;
        a.logC = logC;
        a.tw_lo = d->tw_lo[dir]; a.tw_hi = d->tw_hi[dir]; a.L = d->L;
        a.tw_full = d->tw_full[dir].empty() ? nullptr : d->tw_full[dir][i];
        a.pq = d->pq[dir]; a.pq_shift = d->maxdeg - deg;
        a.pre_mode = (i == 0) ? op.pre_mode : PRE_NONE;
        if (op.pre) { a.pre_lo = op.pre->lo; a.pre_hi = op.pre->hi; }
        a.post_mode = (i == P - 1) ? op.post_mode : POST_NONE;
        a.post_const = op.post_const;
        a.post_full = nullptr;
        if (op.post) { a.post_lo = op.post->lo; a.post_hi = op.post->hi; a.post_full = op.post->full; }
        const uint64_t nblk = ((uint64_t)1 << (d->log_n - deg)) >> logC;
        const size_t lds_bytes = (size_t)2 * sizeof(uint4) << (deg + logC);
        static const int t_lazy = tune("FK_NTT_LAZY", 1);     // 0: butterflies in the canonical form (the round-1 kernel)
        void (*kern)(PassArgs) = t_lazy ? ntt_pass_kernel<FrL> : ntt_pass_kernel<Fr>;
        FK_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        FK_TRY(stats_begin(ctx, ctx->ev_ntt, (uint64_t)1 << d->log_n));
        // one lane per two butterflies (the kernel issues the field products of a pair as dual chains)
        uint32_t threads = 1u << (deg + logC > 1 ? deg + logC - 2 : 0);
        if (threads < 64) threads = 64;
        if (threads > ctx->ntt_threads) threads = ctx->ntt_threads;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(threads), lds_bytes, ctx->stream, a);
        FK_HIP(ctx, hipGetLastError());
        FK_TRY(stats_end(ctx, ctx->ev_ntt));
        src = dst;
        lgp += deg;
    }
    return FK_OK;
}

// In-place transform of a device array (fk_ntt / fk_ntt_dev).
int ntt_exec_simple(fk_ctx *ctx, Fr *d_data, uint32_t log_n, bool inverse, bool coset) {
    NttDomain *d = nullptr;
    FK_TRY(get_domain(ctx, log_n, &d));
    const size_t bytes = sizeof(Fr) << log_n;
    FK_HIP(ctx, ctx->ntt_s1.reserve(bytes));
    FK_HIP(ctx, ctx->ntt_s2.reserve(bytes));
    NttOp op;
    op.inverse = inverse;
    if (!inverse && coset) { op.pre_mode = PRE_TABLE; op.pre = &d->t_g; }            // coset_fft
    if (inverse && !coset) { op.post_mode = POST_CONST; op.post_const = d->minv; }   // ifft
    if (inverse && coset) { op.post_mode = POST_TABLE; op.post = &d->t_ginv_minv; }  // icoset_fft
    if (d->degs.size() >= 2) return ntt_exec(ctx, d, op, d_data, ctx->ntt_s1.as<Fr>(), ctx->ntt_s2.as<Fr>(), d_data);
    FK_TRY(ntt_exec(ctx, d, op, d_data, ctx->ntt_s1.as<Fr>(), ctx->ntt_s2.as<Fr>(), ctx->ntt_s1.as<Fr>()));
    FK_HIP(ctx, hipMemcpyAsync(d_data, ctx->ntt_s1.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FK_OK;
}

// h = (A*B - C)/Z.  d_a, d_b, d_c: device arrays with CAPACITY m = next_pow2(n) elements, the first n
// hold the row evaluations; they are used as scratch.  d_h_out: m elements, the first m-1 are h.
int quotient_dev(fk_ctx *ctx, Fr *d_a, Fr *d_b, Fr *d_c, uint64_t n, Fr *d_h_out, uint64_t *m_out) {
    if (n == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "quotient: n == 0");
    const uint32_t log_n = ceil_log2_u64(n);
    NttDomain *d = nullptr;
    FK_TRY(get_domain(ctx, log_n, &d));
    const uint64_t m = (uint64_t)1 << log_n;
    if (m_out) *m_out = m;
    const size_t bytes = sizeof(Fr) << log_n;
    FK_HIP(ctx, ctx->ntt_s1.reserve(bytes));
    FK_HIP(ctx, ctx->ntt_s2.reserve(bytes));
    Fr *s1 = ctx->ntt_s1.as<Fr>(), *s2 = ctx->ntt_s2.as<Fr>();
    Fr *polys[3] = {d_a, d_b, d_c};
    const bool multi = d->degs.size() >= 2;
    // SIX transforms, not bellman's seven.  bellman: h = icoset_fft((A_c o B_c - C_c) / Z(g)) with X_c = coset_fft(ifft(x)).
    // icoset_fft is linear and undoes coset_fft exactly, so icoset_fft(C_c) is simply ifft(c): c never has to visit the coset,
    //     h_i = [g^-i / (m Z(g))] * ifft(A_c o B_c)_i  -  [1 / Z(g)] * ifft(c)_i
    // -- the same field elements (exact arithmetic: the identity holds for ANY a, b, c, satisfied system or not), one transform less.
    for (int k = 0; k < 3; k++) {
        Fr *x = polys[k];
        if (m > n) FK_HIP(ctx, hipMemsetAsync(x + n, 0, (m - n) * sizeof(Fr), ctx->stream));
        if (k == 2) {       // c: ifft only, with 1 / (m Z(g)) on the way out
            NttOp invc; invc.inverse = true; invc.post_mode = POST_CONST; invc.post_const = Fr::mul(d->minv, d->zinv);
            if (multi) FK_TRY(ntt_exec(ctx, d, invc, x, s1, s2, x));
            else { FK_TRY(ntt_exec(ctx, d, invc, x, s1, s2, s1)); FK_HIP(ctx, hipMemcpyAsync(x, s1, bytes, hipMemcpyDeviceToDevice, ctx->stream)); }
            break;
        }
        // ifft followed by the coset shift g^i (first half of coset_fft), fused: * g^i / m on the way out
        NttOp inv; inv.inverse = true; inv.post_mode = POST_TABLE; inv.post = &d->t_g_minv;
        NttOp fwd;  // the transform part of coset_fft
        if (multi) {
            FK_TRY(ntt_exec(ctx, d, inv, x, s1, s2, x));
            FK_TRY(ntt_exec(ctx, d, fwd, x, s1, s2, x));
        } else {
            FK_TRY(ntt_exec(ctx, d, inv, x, s1, s2, s1));
            FK_TRY(ntt_exec(ctx, d, fwd, s1, s2, s2, x));
        }
    }
    // a*b on the coset (fused into the first pass); g^-i / (m Z(g)) and the subtraction of c's scaled coefficients (fused into the last)
    NttOp fin; fin.inverse = true; fin.pre_mode = PRE_AB; fin.xb = d_b; fin.xc = d_c;
    fin.post_mode = POST_TABLE_SUB; fin.post = &d->t_ginv_minv_zinv;
    FK_TRY(ntt_exec(ctx, d, fin, d_a, s1, s2, d_h_out));
    return FK_OK;
}

// ------------------------------------------------------------------------------------------ distributed quotient
// One process per GPU, W = 2^w ranks, m = W * L.  The transform of size m is cut ONCE, between ranks, so that every
// transform costs one all-to-all of the caller (RCCL over xGMI) and L-point transforms that never leave a GPU:
//
//   variant 1 (cyclic in -> block-cyclic out), j = j1 + W j2, k = k2 + L k1, rank = j1:
//       X[k2 + L k1] = sum_j1 w_W^(j1 k1) * [ w_m^(j1 k2) * sum_j2 x[j1 + W j2] w_L^(j2 k2) ]
//       local L-point transform + twiddle (fk_dq_local_dev) | all-to-all | W-point transform across the received
//       chunks (fk_dq_cross_dev).  Rank p ends up with k2 in [p L/W, (p+1) L/W) for every k1.
//   variant 2 (block-cyclic in -> cyclic out), j = j2 + L j1, k = k1 + W k2:
//       X[k1 + W k2] = sum_j2 w_L^(j2 k2) * w_m^(j2 k1) * sum_j1 x[j2 + L j1] w_W^(j1 k1)
//       W-point transform across chunks + twiddle (fk_dq_cross_dev) | all-to-all | local L-point transform.
//
// The quotient chains them: ifft (v1) . coset_fft (v2) . a*b-c . icoset_fft (v1); the W-point halves of the first two
// meet in ONE kernel (inverse W-point, * g^i / m, forward W-point, twiddle).  A last all-to-all turns the
// block-cyclic coefficients into contiguous blocks -- the h-base sharding of the key (h_slice in common.hpp).
// All scale factors of the single-GPU pipeline (g^i, 1/m, 1/Z(g)) are applied once, in the cross kernels, from the
// m-domain's two-level tables.

__global__ void dq_gather_kernel(const Fr *full, uint64_t n, uint32_t log_w, uint32_t rank, uint64_t L, Fr *local) {
    const uint64_t j2 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j2 >= L) return;
    const uint64_t idx = rank + (j2 << log_w);
    local[j2] = idx < n ? full[idx] : Fr::zero();
}

struct CrossArgs {
    Fr *buf;
    uint32_t log_m, log_w, rank;
    int mode;                              // 0: ifft tail + coset shift + coset_fft head, 1: icoset_fft tail (- sub), 2: ifft tail * post_const
    const Fr *sub;                         // mode 1: subtracted from the result, same block-cyclic layout (c's scaled coefficients)
    Fr post_const;                         // mode 2
    Fr w_fwd[4], w_inv[4];                 // w_W^e and w_W^-e, e < W/2
    const Fr *sc_lo, *sc_hi, *tw_lo, *tw_hi;
    uint32_t Lbits;
};

// W-point decimation-in-frequency transform held in registers, natural order in and out
template <int LOGW>
static __device__ __forceinline__ void small_dft(Fr (&v)[1 << LOGW], const Fr *wt) {
    constexpr int W = 1 << LOGW;
#pragma unroll
    for (int s = 0; s < LOGW; s++) {
        const int half = W >> (s + 1);
#pragma unroll
        for (int blk = 0; blk < W; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                Fr u = v[blk + j], t = v[blk + j + half];
                Fr d;
                Fr::addsub2(u, t, u, t, v[blk + j], d);
                v[blk + j + half] = j ? Fr::mul(d, wt[j << s]) : d;
            }
        }
    }
    Fr o[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < LOGW; b++) r |= ((i >> b) & 1) << (LOGW - 1 - b);
        o[r] = v[i];
    }
#pragma unroll
    for (int i = 0; i < W; i++) v[i] = o[i];
}

template <int LOGW>
__global__ __launch_bounds__(256) void dq_cross_kernel(CrossArgs a) {
    constexpr int W = 1 << LOGW;
    const uint64_t Lc = (uint64_t)1 << (a.log_m - 2 * a.log_w), L = (uint64_t)1 << (a.log_m - a.log_w);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Lc) return;
    const uint32_t mask = (1u << a.Lbits) - 1;
    Fr v[W];
#pragma unroll
    for (int j = 0; j < W; j++) v[j] = a.buf[(uint64_t)j * Lc + t];
    small_dft<LOGW>(v, a.w_inv);
    const uint64_t k2 = (uint64_t)a.rank * Lc + t;
#pragma unroll
    for (int k1 = 0; k1 < W; k1++) {
        const uint64_t i = k2 + (uint64_t)k1 * L;                  // coefficient index
        if (a.mode == 2) v[k1] = Fr::mul(v[k1], a.post_const);
        else v[k1] = Fr::mul(v[k1], Fr::mul(a.sc_lo[i & mask], a.sc_hi[i >> a.Lbits]));
        if (a.mode == 1 && a.sub) v[k1] = Fr::sub(v[k1], a.sub[(uint64_t)k1 * Lc + t]);
    }
    if (a.mode == 0) {
        small_dft<LOGW>(v, a.w_fwd);
#pragma unroll
        for (int k1 = 1; k1 < W; k1++) {
            const uint64_t e = k2 * (uint64_t)k1;                   // < m
            if (e) v[k1] = Fr::mul(v[k1], Fr::mul(a.tw_lo[e & mask], a.tw_hi[e >> a.Lbits]));
        }
    }
#pragma unroll
    for (int j = 0; j < W; j++) a.buf[(uint64_t)j * Lc + t] = v[j];
}

static int dq_check(fk_ctx *ctx, uint32_t log_m, uint32_t rank, uint32_t log_w) {
    if (log_w > 3) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: at most 8 ranks (log_w = %u)", log_w);
    if (rank >> log_w) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: rank %u of %u", rank, 1u << log_w);
    if (log_m < 2 * log_w) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: domain 2^%u too small for %u ranks", log_m, 1u << log_w);
    return FK_OK;
}

int dq_gather(fk_ctx *ctx, const Fr *d_full, uint64_t n, uint32_t log_m, uint32_t rank, uint32_t log_w, Fr *d_local) {
    FK_TRY(dq_check(ctx, log_m, rank, log_w));
    if (n > ((uint64_t)1 << log_m)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: n exceeds the domain");
    const uint64_t L = (uint64_t)1 << (log_m - log_w);
    hipLaunchKernelGGL(dq_gather_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, d_full, n, log_w, rank, L, d_local);
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}

// stage 0: inverse L-point transform + twiddle (ifft, first half)      stage 1: forward L-point transform
// stage 2: x := x * xb - xc, then as stage 0 (icoset_fft, first half)  (coset_fft, second half)
// stage 3: x := x * xb, then as stage 0 (six-transform form: c is subtracted in coefficient space, dq_cross mode 1 with `sub`)
int dq_local(fk_ctx *ctx, Fr *d_x, const Fr *d_xb, const Fr *d_xc, uint32_t log_m, uint32_t rank, uint32_t log_w, int stage) {
    FK_TRY(dq_check(ctx, log_m, rank, log_w));
    if (stage < 0 || stage > 3 || (stage == 2 && (!d_xb || !d_xc)) || (stage == 3 && !d_xb)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: bad stage");
    NttDomain *dl = nullptr, *dm = nullptr;
    FK_TRY(get_domain(ctx, log_m - log_w, &dl));
    FK_TRY(get_domain(ctx, log_m, &dm));
    const size_t bytes = sizeof(Fr) << (log_m - log_w);
    FK_HIP(ctx, ctx->ntt_s1.reserve(bytes));
    FK_HIP(ctx, ctx->ntt_s2.reserve(bytes));
    NttOp op;
    op.inverse = stage != 1;
    if (stage == 2) { op.pre_mode = PRE_ABC; op.xb = d_xb; op.xc = d_xc; }
    if (stage == 3) { op.pre_mode = PRE_AB; op.xb = d_xb; }
    if (stage != 1 && rank != 0) {
        const uint64_t key = ((uint64_t)log_w << 32) | rank;
        auto it = dl->dist_post.find(key);
        if (it == dl->dist_post.end()) {
            ScaleTable t;
            FK_TRY(make_scale_table(ctx, dl, Fr::pow_u64(dm->omega_inv, rank), Fr::one(), &t));
            it = dl->dist_post.emplace(key, t).first;
        }
        op.post_mode = POST_TABLE; op.post = &it->second;
    }
    Fr *s1 = ctx->ntt_s1.as<Fr>(), *s2 = ctx->ntt_s2.as<Fr>();
    if (dl->degs.size() >= 2) return ntt_exec(ctx, dl, op, d_x, s1, s2, d_x);
    FK_TRY(ntt_exec(ctx, dl, op, d_x, s1, s2, s1));
    FK_HIP(ctx, hipMemcpyAsync(d_x, s1, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FK_OK;
}

int dq_cross(fk_ctx *ctx, Fr *d_buf, uint32_t log_m, uint32_t rank, uint32_t log_w, int mode, const Fr *d_sub) {
    FK_TRY(dq_check(ctx, log_m, rank, log_w));
    if (mode < 0 || mode > 2 || (d_sub && mode != 1)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "distributed quotient: bad mode");
    NttDomain *dm = nullptr;
    FK_TRY(get_domain(ctx, log_m, &dm));
    CrossArgs a{};
    a.buf = d_buf; a.log_m = log_m; a.log_w = log_w; a.rank = rank; a.mode = mode; a.sub = d_sub;
    a.post_const = Fr::mul(dm->minv, dm->zinv);
    const uint64_t L = (uint64_t)1 << (log_m - log_w);
    const Fr ww = Fr::pow_u64(dm->omega, L), wwi = Fr::pow_u64(dm->omega_inv, L);     // w_W, w_W^-1
    Fr cf = Fr::one(), ci = Fr::one();
    for (int e = 0; e < 4; e++) { a.w_fwd[e] = cf; a.w_inv[e] = ci; cf = Fr::mul(cf, ww); ci = Fr::mul(ci, wwi); }
    const ScaleTable &sc = mode == 0 ? dm->t_g_minv : dm->t_ginv_minv_zinv;
    a.sc_lo = sc.lo; a.sc_hi = sc.hi; a.tw_lo = dm->tw_lo[0]; a.tw_hi = dm->tw_hi[0]; a.Lbits = dm->L;
    const uint64_t Lc = (uint64_t)1 << (log_m - 2 * log_w);
    const dim3 grid((unsigned)((Lc + 255) / 256)), block(256);
    switch (log_w) {
        case 0: hipLaunchKernelGGL(dq_cross_kernel<0>, grid, block, 0, ctx->stream, a); break;
        case 1: hipLaunchKernelGGL(dq_cross_kernel<1>, grid, block, 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(dq_cross_kernel<2>, grid, block, 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(dq_cross_kernel<3>, grid, block, 0, ctx->stream, a); break;
    }
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}

}  // namespace fk
