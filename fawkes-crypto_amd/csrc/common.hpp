// Shared host-side plumbing of libfawkes_hip.so: context, error handling, grow-only device scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>
#include <functional>
#include "../../include/fawkes_hip.h"
#include "curve.hpp"

namespace fk {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // grow-only; contents are NOT preserved across a growth
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, bytes); want = bytes; }      // (the failed attempt must not stay behind as the thread's "last error")
        if (e == hipSuccess) cap = want; else { p = nullptr; (void)hipGetLastError(); }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

struct NttDomain;

struct EventPair { hipEvent_t a, b; uint64_t units; };

// Multi-scalar multiplications run on MSM_LANES independent "lanes" (stream + private scratch) used in turn, so that the
// memory-bound digit sort of multiplication k+1 and the latency-bound overflow / bucket-reduction kernels of
// multiplication k run underneath the VALU-bound bucket accumulation of the other lane (msm.hip).
struct MsmLane {
    hipStream_t st = nullptr;
    hipEvent_t ev_in = nullptr;     // main stream -> lane: the scalars are ready
    hipEvent_t ev_sorted = nullptr;   // the lane's latest sort is complete (the prover gates the quotient on it)
    bool ev_sorted_valid = false;
    DevBuf digits, sorted, totals, starts, perm, overlist, tasktab, partials, s2_cnt1, s2_seg, s2_cnt2, s2_tmp_idx, s2_tmp_lo, buckets;
    DevBuf buckets2;     // the buckets of a multiplication that reuses this lane's sort (B2 after B1): its accumulation is queued right behind B1's, before B1's tail has read `buckets`
    void *h_stage = nullptr;        // pinned host staging: counters read back, oversized-bucket list, task tables
    size_t h_cap = 0;
    const void *last_sort_scalars = nullptr; size_t last_sort_n = 0; unsigned last_sort_c = 0; bool last_merged = false;   // what `sorted` currently holds
    // ... the oversized-bucket state and tables that belong to it live on the device (msm.hip: MsmDyn, tasktab)
};

// One outstanding multiplication: everything is queued, `done` fires when its window sums are in h_wp.
struct MsmTail {
    bool active = false;
    uint32_t cb = 0, wide = 0, W = 0, nblk = 0;     // window widths (msm.hip: MsmPlan), windows, partial sums per window
    hipEvent_t done = nullptr;
    void *h_wp = nullptr;           // pinned host copy of the window partial sums
    size_t h_cap = 0;
    DevBuf d_wp;
};
static constexpr int MSM_TAILS = 12;     // five per proof; a second proof's four witness multiplications may be begun before the first one's are collected (early front)
static constexpr int MSM_LANES = 4;   // B pair, L, A and H each on a lane of their own in the sorts-first schedule (prover.hip)

// Which variables feed the A and the B query, as index lists (they are structural: a resident constraint system
// computes them once at load).  With them the per-proof scalar compaction is one gather per query, no host round trip.
struct QueryIdx {
    const uint8_t *d_a_aux = nullptr, *d_b_in = nullptr, *d_b_aux = nullptr;   // the density maps the lists were made from
    const uint32_t *a = nullptr, *b = nullptr;                                 // variable indices, query order
    uint64_t n_a = 0, n_b = 0;
};

// Fixed-base precomputation of one key array (msm.hip): level w holds 2^(offset of window w) * P for every base P, so that
// the buckets of ALL windows carry the same weights and are accumulated as ONE bucket set (one reduction instead of W).
// lev: levels 1 .. W-1, n points each (level 0 is the key array itself).  Tied to the window plan (cb, wide, W) of n.
struct KeyPre {
    void *lev = nullptr;
    size_t n = 0;
    uint32_t cb = 0, wide = 0, W = 0;
};

}  // namespace fk

struct fk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    unsigned window_bits = 0;  // 0 = auto
    unsigned ntt_threads = 512;  // workgroup size cap of the NTT pass kernel (FK_NTT_THREADS overrides)
    std::map<uint32_t, fk::NttDomain *> domains;
    // MSM: two lanes used in turn, one record per outstanding multiplication
    fk::MsmLane lanes[fk::MSM_LANES];
    int lane_next = 0, lane_prev = 0;
    int co_tenants = 1;                   // contexts of one fk_multi that share this device (ranks on one GPU: tests, rehearsals): each needs its own scratch
    int lanes_in_use = fk::MSM_LANES;   // the provers use 2 for the largest domains (measured, prover.hip)
    fk::MsmTail tails[fk::MSM_TAILS];
    fk::DevBuf misc;
    // witness multiplications (L, A, B1, B2) in flight: begun before / while the quotient runs on the main stream
    hipStream_t aux = nullptr;          // scalar compaction for the A / B queries
    hipEvent_t ev_aux = nullptr, ev_main = nullptr, ev_z = nullptr;
    // sorts-first schedule (prover.hip): with defer_back set, msm_begin queues only the front of a multiplication (digits, sort,
    // size ordering) and leaves the rest (accumulation, oversized buckets, reduction, download) here, to be queued by
    // msm_run_deferred once the caller has put the quotient between the two
    bool defer_back = false;
    std::vector<std::function<int()>> deferred, deferred_tails;     // accumulations; tails
    bool ev_z_recorded = false;
    hipEvent_t ev_acc_done = nullptr; bool ev_acc_done_valid = false;   // behind the most recent bucket accumulation (any lane)
    bool wit_active = false;
    // Early front (pipelined proofs, sorts-first schedule): while proof k's last kernels run -- latency-bound tails that leave the GPU
    // mostly idle -- the memory-bound front of proof k+1 (evaluation of a, b, c and the witness sorts) is already queued behind
    // proof k's last accumulation.  before_block: called by the prover when everything of proof k is queued, before it blocks on
    // the results; early: what that front left for proof k+1's run to pick up.
    std::function<int()> before_block;
    bool gather_on_main = false;
    struct EarlyFront { bool done = false; int tails[4] = {-1, -1, -1, -1}; const fk_key *key = nullptr; const struct fk_r1cs_dev *r1cs = nullptr; const void *d_z = nullptr; } early;
    const fk::QueryIdx *qidx = nullptr;   // set by the resident-constraint-system entry points for the duration of a call
    int wit_tail[4] = {-1, -1, -1, -1}; // B1, B2, L, A
    // witness hand-over from host memory (fk_witness_upload_async / fk_prove_r1cs_submit): two device slots filled on a copy
    // stream of their own, so that the upload of proof k+1's witness runs underneath proof k
    hipStream_t copy_st = nullptr;
    struct WitSlot {
        fk::DevBuf buf; hipEvent_t ready = nullptr; bool pending = false;        // pending: a submitted proof waits in this slot
        hipEvent_t part = nullptr;               // sharded hand-over (fk_witness_upload_part_async): this rank's own piece has arrived; `ready` then follows the all-gather
        // deferred: the upload has not been queued yet -- the proof that runs first queues it behind its memory-bound front
        // (upload_deferred), so that the copy runs underneath transforms and accumulations instead of beside sorts
        bool deferred = false; const void *host_z = nullptr; size_t host_bytes = 0;
        const fk_key *key = nullptr; const struct fk_r1cs_dev *r1cs = nullptr; uint64_t r[4], s[4];
    } wslot[2];
    int wslot_next = 0;
    hipEvent_t ev_upload_gate = nullptr;
    hipEvent_t ev_chunk[16] = {nullptr};      // fk_prove_r1cs, chunked hand-over: piece j of the witness has landed (spmv.hip)
    // NTT / prover scratch
    fk::DevBuf ntt_s1, ntt_s2, ntt_io, hbuf, sc_a, sc_b, scan_tmp, stage_a, stage_b, stage_c, stage_z, stage_d;
    // stats
    std::vector<fk::EventPair> ev_acc, ev_acc2, ev_ntt;   // G1 accumulate, G2 accumulate, NTT passes
    std::vector<hipEvent_t> ev_pool;
    uint64_t acc_adds[2] = {0, 0};        // mixed additions done by the G1 / G2 accumulate kernels since the last reset
    bool stats_on = true;
    bool debug = false;   // FK_DEBUG=1: synchronise and log after every launch
};

struct fk_key {
    uint64_t m = 0;
    uint32_t num_input = 0, num_aux = 0;
    uint64_t n_h = 0, n_l = 0, n_a = 0, n_b = 0;          // full (unsharded) counts
    uint32_t shard_index = 0, shard_count = 1;
    // [lo, hi) slices held by this context
    uint64_t h_lo = 0, h_hi = 0, l_lo = 0, l_hi = 0, a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;      // b_*: the slice of b_g1
    uint64_t b2_lo = 0, b2_hi = 0;                       // the slice of b_g2 (equal to b_g1's unless the key is split by work, FK_Z_WORK_SPLIT)
    fk::G1Affine *d_h = nullptr, *d_l = nullptr, *d_a = nullptr, *d_b1 = nullptr;
    fk::G2Affine *d_b2 = nullptr;
    fk::G1Affine alpha_g1, beta_g1, delta_g1;
    fk::G2Affine beta_g2, delta_g2;
    fk::KeyPre pre_h, pre_l, pre_a, pre_b1, pre_b2;       // optional (memory permitting), see key_precompute
    double load_s[2] = {0, 0};                            // seconds the loader spent on the arrays (transfer, conversion, checks / derivation) and on the fixed-base levels
};

#define FK_SET_ERR(ctx, code, ...)                                   \
    do {                                                             \
        char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__);          \
        (ctx)->err = _b;                                             \
        return (code);                                               \
    } while (0)

#define FK_HIP(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            (ctx)->err = _b;                                                                      \
            (void)hipGetLastError();   /* reported: do not leave it as the thread's sticky "last error" for the next call's check */ \
            return (_e == hipErrorOutOfMemory) ? FK_ERR_OOM : FK_ERR_HIP;                         \
        }                                                                                         \
    } while (0)

// debug aid: FK_DEBUG=1 makes every stage synchronise and report (stderr)
#define FK_DBG(ctx, name)                                                                         \
    do {                                                                                          \
        if ((ctx)->debug) {                                                                       \
            fprintf(stderr, "[fk] launch %s ...", name); fflush(stderr);                          \
            hipError_t _e = hipStreamSynchronize((ctx)->stream);                                  \
            fprintf(stderr, " %s\n", hipGetErrorString(_e)); fflush(stderr);                      \
        }                                                                                         \
    } while (0)

#define FK_TRY(expr) do { int _rc = (expr); if (_rc != FK_OK) return _rc; } while (0)
#define FK_CAT2_(a, b) a##b
#define FK_CAT_(a, b) FK_CAT2_(a, b)
#define FK_RANGE(name) fk::RoctxRange FK_CAT_(fk_range_, __LINE__)(name)

// Every extern "C" entry that can allocate on the host (std::vector / std::string / std::function behind almost all of them) runs
// its body through this guard: a C++ exception must never leave an extern "C" function -- std::terminate would take the Rust or
// ctypes host down, where the reference's `prove` at worst panics ("Errors": status codes, never aborts).  Out of host
// memory is FK_ERR_OOM like out of device memory; anything else FK_ERR_HIP with what() as the message.  `c` is the fk_ctx / fk_multi
// that carries the error string (may be null: code only).
template <class C, class F>
static inline int fk_guard(C *c, F &&body) noexcept {
    auto note = [&](const char *msg) noexcept { if (c) { try { c->err = msg; } catch (...) {} } };
    try { return body(); }
    catch (const std::bad_alloc &) { note("out of host memory"); return FK_ERR_OOM; }
    catch (const std::length_error &) { note("out of host memory (a container larger than this host can hold)"); return FK_ERR_OOM; }
    catch (const std::exception &e) { note(e.what()); return FK_ERR_HIP; }
    catch (...) { note("unknown C++ exception"); return FK_ERR_HIP; }
}

namespace fk {

std::string &tls_error();      // gatestream.hip: what fk_last_error(NULL) returns (context-free calls leave their message here)
unsigned host_threads();       // gatestream.hip: FK_HOST_THREADS, else the cores this process may use
bool cu_masks(::fk_ctx *ctx, std::vector<uint32_t> &compute, std::vector<uint32_t> &mem);      // msm.hip: FK_CU_SPLIT (experiment builds)

// roctx ranges around the library's phases (tracing).  FK_ROCTX=1 binds libroctx64.so.4 with dlopen on first use; a trace taken with
// `rocprofv3 --marker-trace --kernel-trace -- python3 bench.py ...` then shows which host call queued which kernels (the proof's kernels run
// asynchronously: a range brackets the QUEUEING of a phase and, where the call blocks, the wait).  Unset: one predictable branch per range.
struct RoctxApi { int (*push)(const char *) = nullptr; int (*pop)() = nullptr; };
const RoctxApi &roctx_api();   // gatestream.hip
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) { const RoctxApi &r = roctx_api(); on = r.push != nullptr; if (on) (void)r.push(name); }
    ~RoctxRange() { if (on) (void)roctx_api().pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

// Tuning knobs.  A release build compiles the measured default in; `make EXP=1` (-DFK_EXPERIMENTS, libfawkes_hip_exp.so, loaded
// with FK_LIB_VARIANT=exp) reads them from the environment for same-box A/B runs.  The run-time switches of a release build
// are few and documented in DESIGN.md: FK_DEBUG, FK_MSM_PRECOMP, FK_MSM_PRE_MIN_LOG2, FK_PROVE_SORTS_FIRST, FK_SPMV_BIN_MIN,
// FK_OVERLAP_WITNESS, FK_MULTI_HOST_EVENTS.
static inline int tune(const char *name, int dflt) {
#ifdef FK_EXPERIMENTS
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

static inline uint32_t ceil_log2_u64(uint64_t n) { uint32_t k = 0; while (((uint64_t)1 << k) < n) k++; return k; }

// The h bases are sharded in blocks of the evaluation domain (m = n_h + 1 slots, the last one clipped): shard g holds
// h[g*m/W, (g+1)*m/W) -- exactly the block of quotient coefficients the distributed quotient (ntt.hip, fk_dq_*)
// leaves on rank g, so no element has to move before the H multi-scalar multiplication.
static inline void h_slice(uint64_t n_h, uint32_t idx, uint32_t cnt, uint64_t *lo, uint64_t *hi) {
    const uint64_t m = n_h + 1;
    uint64_t a = (uint64_t)((unsigned __int128)m * idx / cnt), b = (uint64_t)((unsigned __int128)m * (idx + 1) / cnt);
    if (a > n_h) a = n_h;
    if (b > n_h) b = n_h;
    *lo = a; *hi = b;
}

// FK_Z_WORK_SPLIT: the four witness arrays laid end to end on a line measured in WORK (a G2 point costs FK_G2_WORK G1 points: the
// accumulation of a 128-byte point is ~2.8 G1 additions), the line cut into `count` equal pieces; out = [lo, hi) of l, a, b_g1, b_g2
// for piece `index`.  A rank then holds one or two LARGE pieces (e.g. 19.4 M G1 points of l) instead of an eighth of each of the
// four arrays: fewer, larger multiplications per rank (the fixed sort set-up and reduction latency of a multiplication does not
// shrink with its size, and the accumulation of a 4 M-point shard runs at 75 - 85 % of its full-size efficiency, DESIGN.md section
// 4.4).  h stays in blocks of the domain (what the distributed quotient leaves on the rank).  The cuts are monotone per array, so
// the pieces tile every array exactly.
// handicap: fixed work of piece 0 in the line's units (FK_Z_WORK_SPLIT_Q0: the evaluation, the whole quotient and H on shard 0); the line is
// then total + handicap long, cut into equal pieces, and piece 0's part of the ARRAYS is what is left of its piece (possibly nothing).
static inline void work_slices(uint64_t n_l, uint64_t n_a, uint64_t n_b, uint32_t index, uint32_t count, uint64_t out[8], long double handicap = 0.0L) {
    const long double seg_w[4] = {1.0L, 1.0L, 1.0L, (long double)FK_G2_WORK};
    const uint64_t seg_n[4] = {n_l, n_a, n_b, n_b};
    long double total = 0; for (int i = 0; i < 4; i++) total += seg_w[i] * (long double)seg_n[i];
    auto cut = [&](uint32_t g, int seg) -> uint64_t {           // index in array `seg` of the g-th cut of the line
        if (g >= count) return seg_n[seg];
        if (g == 0) return 0;
        long double x;
        if (handicap > 0 && count > 1 && handicap * (long double)count >= total + handicap)
            x = total * (long double)(g - 1) / (long double)(count - 1);      // piece 0's fixed work alone fills its share: the arrays go to the others, equally
        else
            x = (total + handicap) * (long double)g / (long double)count - handicap;
        long double s0 = 0; for (int i = 0; i < seg; i++) s0 += seg_w[i] * (long double)seg_n[i];
        if (x <= s0) return 0;
        const long double q = (x - s0) / seg_w[seg];
        return q >= (long double)seg_n[seg] ? seg_n[seg] : (uint64_t)q;
    };
    for (int seg = 0; seg < 4; seg++) { out[2 * seg] = cut(index, seg); out[2 * seg + 1] = cut(index + 1, seg); }
}

// Which part of l, a, b_g1, b_g2 a key holds (one rule for fk_key_load, fk_key_load_bellman, fk_setup*, fk_key_synthetic).
// zlo < 0 (FK_Z_EQUAL_SPLIT): the equal split [index/count, (index+1)/count).  Otherwise the fractions [zlo, zhi) of
// every array -- and (0, 0) then IS the empty slice (the rank that computes the quotient may hold no witness points at
// all).  A lone shard must hold everything: a zero-initialised descriptor is refused instead of proving from nothing.
static inline int key_plan_slices(fk_ctx *ctx, fk_key *k, double zlo, double zhi) {
    h_slice(k->n_h, k->shard_index, k->shard_count, &k->h_lo, &k->h_hi);
    auto eq = [&](uint64_t n, uint64_t *lo, uint64_t *hi) {
        *lo = (uint64_t)((unsigned __int128)n * k->shard_index / k->shard_count);
        *hi = (uint64_t)((unsigned __int128)n * (k->shard_index + 1) / k->shard_count);
    };
    auto fr = [&](uint64_t n, uint64_t *olo, uint64_t *ohi) {
        uint64_t a = (uint64_t)((long double)n * zlo + 0.5L), b = zhi >= 1.0 ? n : (uint64_t)((long double)n * zhi + 0.5L);
        if (a > n) a = n;
        if (b > n) b = n;
        if (b < a) b = a;
        *olo = a; *ohi = b;
    };
    if (zlo <= -1.5) {         // FK_Z_WORK_SPLIT, FK_Z_WORK_SPLIT_Q0
        uint64_t r[8];
        const bool q0 = zlo <= -2.5 && k->shard_count > 1;
        if (q0) { k->h_lo = k->shard_index == 0 ? 0 : k->n_h; k->h_hi = k->n_h; }           // all of h on shard 0, none elsewhere
        work_slices(k->n_l, k->n_a, k->n_b, k->shard_index, k->shard_count, r, q0 ? (long double)FK_Q0_HANDICAP * (long double)k->m : 0.0L);
        k->l_lo = r[0]; k->l_hi = r[1]; k->a_lo = r[2]; k->a_hi = r[3]; k->b_lo = r[4]; k->b_hi = r[5]; k->b2_lo = r[6]; k->b2_hi = r[7];
        return FK_OK;
    }
    if (zlo < 0.0) { eq(k->n_l, &k->l_lo, &k->l_hi); eq(k->n_a, &k->a_lo, &k->a_hi); eq(k->n_b, &k->b_lo, &k->b_hi); k->b2_lo = k->b_lo; k->b2_hi = k->b_hi; return FK_OK; }
    if (!(zhi <= 1.0 && zlo <= zhi)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: bad z fraction range [%g, %g)", zlo, zhi);
    if (k->shard_count == 1 && !(zlo == 0.0 && zhi >= 1.0))
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: a single shard holds the whole arrays (z_frac_lo = FK_Z_EQUAL_SPLIT or [0, 1)), got [%g, %g)", zlo, zhi);
    fr(k->n_l, &k->l_lo, &k->l_hi); fr(k->n_a, &k->a_lo, &k->a_hi); fr(k->n_b, &k->b_lo, &k->b_hi);
    k->b2_lo = k->b_lo; k->b2_hi = k->b_hi;
    return FK_OK;
}

// stats helpers (HIP events on the library stream)
int stats_begin(fk_ctx *ctx, std::vector<EventPair> &v, uint64_t units, hipStream_t st = nullptr);   // nullptr: ctx->stream
int stats_end(fk_ctx *ctx, std::vector<EventPair> &v, hipStream_t st = nullptr);

// ntt.hip
int ntt_exec_simple(fk_ctx *ctx, Fr *d_data, uint32_t log_n, bool inverse, bool coset);
int quotient_dev(fk_ctx *ctx, Fr *d_a, Fr *d_b, Fr *d_c, uint64_t n, Fr *d_h_out, uint64_t *m_out);
void ntt_free_domains(fk_ctx *ctx);
int fr_mul_batch_dev(fk_ctx *ctx, const Fr *a, const Fr *b, Fr *o, size_t n);
int dq_gather(fk_ctx *ctx, const Fr *d_full, uint64_t n, uint32_t log_m, uint32_t rank, uint32_t log_w, Fr *d_local);
int dq_local(fk_ctx *ctx, Fr *d_x, const Fr *d_xb, const Fr *d_xc, uint32_t log_m, uint32_t rank, uint32_t log_w, int stage);
int dq_cross(fk_ctx *ctx, Fr *d_buf, uint32_t log_m, uint32_t rank, uint32_t log_w, int mode, const Fr *d_sub = nullptr);

// msm.hip
int msm_g1_dev(fk_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, G1Xyzz *out, const KeyPre *pre = nullptr);
// split form: *_begin queues the whole multiplication on one of the two lanes and returns a tail handle (-1 for an
// empty sum); *_end waits for it and folds the window sums on the host.  Several multiplications may be outstanding;
// msm_abandon drops them all (error paths); msm_sync waits for both lanes.
// ready: event after which bases / scalars are valid (nullptr: everything queued on ctx->stream so far)
// pre: the bases' precomputed levels (nullptr / empty / made for another n: the ordinary W-bucket-set path)
int msm_g1_begin(fk_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, int *tail, hipEvent_t ready = nullptr, const KeyPre *pre = nullptr);
int msm_g1_end(fk_ctx *ctx, int tail, G1Xyzz *out);
int msm_g2_begin(fk_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, bool reuse_sort, int *tail, hipEvent_t ready = nullptr, const KeyPre *pre = nullptr);
// fills key->pre_* for the slices the key holds (FK_MSM_PRECOMP=0 disables; skipped silently when HBM is short); key_pre_free undoes
int key_precompute(fk_ctx *ctx, fk_key *key);
int key_levels_headroom(fk_ctx *ctx, const fk_key *key, int64_t *bytes);
void key_pre_free(fk_key *key);
int msm_g2_end(fk_ctx *ctx, int tail, G2Xyzz *out);
void msm_abandon(fk_ctx *ctx);
int msm_run_deferred(fk_ctx *ctx, hipEvent_t after);
int upload_deferred(fk_ctx *ctx, bool gate_on_main);      // queues the witness uploads fk_prove_r1cs_submit left for later
int witness_slot_reserve(fk_ctx *ctx, int slot, size_t bytes, bool *moved = nullptr);      // prover.hip: room for `bytes` in a witness slot (+ its stream and events)
int early_witness_begin(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux, const void *d_b_in, const void *d_b_aux, int tails_out[4]);   // prover.hip
bool early_front_applies(const fk_key *key);
void msm_release(fk_ctx *ctx);
int streams_init(fk_ctx *ctx);       // msm.hip: the context's copy / auxiliary / lane streams, created together in a fixed order (fk_init)
int msm_sync(fk_ctx *ctx);
// reuse_sort: the scalars are the ones of the immediately preceding MSM call on this context (same pointer
// and n), so its digits / bucket sort are still valid and are not recomputed (B1 and B2 share scalars)
int msm_g2_dev(fk_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, G2Xyzz *out, bool reuse_sort = false, const KeyPre *pre = nullptr);
int gen_points_g1(fk_ctx *ctx, G1Affine *d_out, size_t n, uint64_t seed);
int gen_points_g2(fk_ctx *ctx, G2Affine *d_out, size_t n, uint64_t seed);
int gen_scalars(fk_ctx *ctx, Fr *d_out, size_t n, uint64_t seed, int kind);
// out[k] = z[j] for the k-th j with density[j] != 0 (device pointers); returns count via *n_out
int gather_scalars(fk_ctx *ctx, const Fr *d_z, const uint32_t *d_idx, size_t n, Fr *d_out, hipStream_t st);
int compact_scalars(fk_ctx *ctx, const Fr *d_z, const uint8_t *d_density, size_t n, Fr *d_out, uint64_t *n_out, hipStream_t st = nullptr);

}  // namespace fk
