// C-ABI entry points of libfawkes_hip.so (declared in include/fawkes_hip.h) and the Groth16 prover
// pipeline that replaces bellman's `create_proof` behind
// /root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80 (algorithm: SURVEY.md App. A.1-A.5).
//
// Device pipeline per proof (one HIP stream, key resident in HBM):
//   quotient (6 fused NTTs for bellman's 7, ntt.hip)  ->  h
//   scalar vectors: h | z_aux | z_in ++ compact(z_aux, a_aux) | compact(z_in, b_in) ++ compact(z_aux, b_aux)
//   five Pippenger MSMs (msm.hip) against the resident key slices
//   host: XYZZ -> affine, proof assembly with r, s and the vk points (a handful of group operations)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "common.hpp"
#include <chrono>
#include <mutex>
#include <string.h>
#include <stdlib.h>

using namespace fk;

namespace fk {

int stats_begin(fk_ctx *ctx, std::vector<EventPair> &v, uint64_t units, hipStream_t st) {
    if (!ctx->stats_on) return FK_OK;
    if (v.size() >= 4096) {      // nobody is reading the statistics: recycle the events instead of growing without bound
        for (auto &old : v) { ctx->ev_pool.push_back(old.a); ctx->ev_pool.push_back(old.b); }
        v.clear();
    }
    EventPair ep{};
    ep.units = units;
    for (hipEvent_t *e : {&ep.a, &ep.b}) {
        if (!ctx->ev_pool.empty()) { *e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
        else FK_HIP(ctx, hipEventCreate(e));
    }
    FK_HIP(ctx, hipEventRecord(ep.a, st ? st : ctx->stream));
    v.push_back(ep);
    return FK_OK;
}
int stats_end(fk_ctx *ctx, std::vector<EventPair> &v, hipStream_t st) {
    if (!ctx->stats_on) return FK_OK;
    FK_HIP(ctx, hipEventRecord(v.back().b, st ? st : ctx->stream));
    return FK_OK;
}

static void g1_to_raw(uint8_t out[64], const G1Xyzz &p) { G1Affine a = p.to_affine(); memcpy(out, &a, 64); }
static void g2_to_raw(uint8_t out[128], const G2Xyzz &p) { G2Affine a = p.to_affine(); memcpy(out, &a, 128); }
static G1Affine g1_from_raw(const uint8_t *b) { G1Affine a; memcpy(&a, b, 64); return a; }
static G2Affine g2_from_raw(const uint8_t *b) { G2Affine a; memcpy(&a, b, 128); return a; }

static void slice(uint64_t n, uint32_t idx, uint32_t cnt, uint64_t *lo, uint64_t *hi) {
    *lo = (uint64_t)((unsigned __int128)n * idx / cnt);
    *hi = (uint64_t)((unsigned __int128)n * (idx + 1) / cnt);
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace fk

static_assert(sizeof(G1Affine) == 64 && sizeof(G2Affine) == 128 && sizeof(Fr) == 32, "raw layouts");

extern "C" {

// ------------------------------------------------------------------------------------------ context
int fk_init(int device_id, fk_ctx **out) {
    if (!out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FK_ERR_HIP;   // no GPU: fail loudly, no CPU fallback
    if (device_id < 0 || device_id >= ndev) return FK_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return FK_ERR_HIP;
    fk_ctx *ctx = new fk_ctx();
    ctx->device = device_id;
    { const char *d = getenv("FK_DEBUG"); ctx->debug = d && d[0] && d[0] != '0'; }
    // several PROCESSES proving on this GPU (each with a context of its own): the key loaders reserve what a proof allocates later that many times
    { const char *t = getenv("FK_CO_TENANTS"); if (t) { const int v = atoi(t); if (v >= 1 && v <= 64) ctx->co_tenants = v; } }
    if (ctx->debug || getenv("FK_BACKTRACE")) {
        // debugging aid (FK_DEBUG=1 or FK_BACKTRACE=1): a native backtrace on SIGSEGV / SIGABRT (the Python host's faulthandler only shows Python frames; the
        // offsets resolve against the same libfawkes_hip.so with addr2line / llvm-symbolizer)
        // Installed once (std::call_once), with sigaction, CHAINING to whatever handler the host had (Python's faulthandler, Rust's
        // stack-overflow guard): ours prints and hands the signal on.  backtrace() is called once here, at install time, so that its
        // lazy initialisation (it may dlopen libgcc and allocate) does not happen inside a signal handler.
        static std::once_flag once;
        std::call_once(once, [] {
            void *warm[4];
            (void)backtrace(warm, 4);
            static struct sigaction prev_segv, prev_abrt;
            auto h = +[](int sig, siginfo_t *info, void *uctx) {
                void *bt[64];
                const int n = backtrace(bt, 64);
                const char msg[] = "[fk] fatal signal, native backtrace:\n";
                (void)!write(2, msg, sizeof msg - 1);
                backtrace_symbols_fd(bt, n, 2);
                const struct sigaction &prev = sig == SIGSEGV ? prev_segv : prev_abrt;
                if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) { prev.sa_sigaction(sig, info, uctx); return; }
                if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler) { prev.sa_handler(sig); return; }
                signal(sig, SIG_DFL); raise(sig);
            };
            struct sigaction sa;
            memset(&sa, 0, sizeof sa);
            sa.sa_sigaction = h;
            sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
            sigemptyset(&sa.sa_mask);
            sigaction(SIGSEGV, &sa, &prev_segv);
            sigaction(SIGABRT, &sa, &prev_abrt);
        });
    }
    { const int v = tune("FK_NTT_THREADS", 512); if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) ctx->ntt_threads = (unsigned)v; }
    {   // FK_CU_SPLIT_ALL=1 (experiment builds): a GUEST context -- every stream, the main one included, on the set-aside units of FK_CU_SPLIT
        std::vector<uint32_t> mc, mm;
        if (tune("FK_CU_SPLIT_ALL", 0) && cu_masks(ctx, mc, mm)) {
            if (hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)mm.size(), mm.data()) != hipSuccess) { delete ctx; return FK_ERR_HIP; }
        } else if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return FK_ERR_HIP; }
    }
    { const char *e = getenv("FK_LAZY_STREAMS");       // (=1: the streams are created on first use, as before round 5 -- for the A/B only)
      if (!(e && e[0] == '1') && streams_init(ctx) != FK_OK) { fk_free(ctx); return FK_ERR_HIP; } }
    *out = ctx;
    return FK_OK;
}

void fk_free(fk_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ntt_free_domains(ctx);
    msm_release(ctx);
    for (DevBuf *b : {&ctx->misc, &ctx->ntt_s1, &ctx->ntt_s2, &ctx->ntt_io,
                      &ctx->hbuf, &ctx->sc_a, &ctx->sc_b, &ctx->scan_tmp, &ctx->stage_a, &ctx->stage_b, &ctx->stage_c,
                      &ctx->stage_z, &ctx->stage_d})
        b->release();
    for (auto &v : {&ctx->ev_acc, &ctx->ev_acc2, &ctx->ev_ntt}) for (auto &ep : *v) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->copy_st) { (void)hipStreamSynchronize(ctx->copy_st); (void)hipStreamDestroy(ctx->copy_st); }
    for (auto &w : ctx->wslot) { w.buf.release(); if (w.ready) (void)hipEventDestroy(w.ready); if (w.part) (void)hipEventDestroy(w.part); }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ctx == NULL: the message of the calling thread's latest context-free call (fk_gates_decode / fk_gates_encode / fk_verify with a NULL context)
const char *fk_last_error(const fk_ctx *ctx) { return ctx ? ctx->err.c_str() : fk::tls_error().c_str(); }

// Releases everything the context has grown for the proofs it has run -- the MSM lanes' scratch, the transforms' tables and buffers, the
// staging vectors, the two witness slots -- and keeps keys and resident constraint systems.  All of it is re-allocated on demand.
// A service that moves to a larger key calls this first: the scratch of a 2^27 proof (~170 GB) would otherwise stand in the way of the
// next key's fixed-base levels (the loader then skips the levels that do not fit, with a warning -- never silently).
int fk_trim(fk_ctx *ctx) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    for (const auto &w : ctx->wslot) if (w.pending) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "trim: a submitted proof is outstanding (call fk_prove_r1cs_wait first)");
    if (ctx->wit_active || ctx->early.done || !ctx->deferred.empty()) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "trim: multiplications are in flight");
    for (const auto &t : ctx->tails) if (t.active) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "trim: a multiplication begun with a *_begin_* call has not been collected");
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FK_TRY(msm_sync(ctx));
    if (ctx->copy_st) FK_HIP(ctx, hipStreamSynchronize(ctx->copy_st));
    ntt_free_domains(ctx);
    msm_release(ctx);
    for (DevBuf *b : {&ctx->misc, &ctx->ntt_s1, &ctx->ntt_s2, &ctx->ntt_io, &ctx->hbuf, &ctx->sc_a, &ctx->sc_b, &ctx->scan_tmp, &ctx->stage_a, &ctx->stage_b,
                      &ctx->stage_c, &ctx->stage_z, &ctx->stage_d})
        b->release();
    for (auto &w : ctx->wslot) { w.buf.release(); w.deferred = false; if (w.ready) { (void)hipEventDestroy(w.ready); w.ready = nullptr; } if (w.part) { (void)hipEventDestroy(w.part); w.part = nullptr; } }      // "holds nothing" again
    ctx->lane_prev = 0; ctx->lane_next = 0;
    { const char *e = getenv("FK_LAZY_STREAMS"); if (!(e && e[0] == '1')) FK_TRY(streams_init(ctx)); }      // the streams come back together, in the same order
    return FK_OK;
}); }

int fk_set_window_bits(fk_ctx *ctx, unsigned c) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (c != 0 && (c < 2 || c > 22)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "window bits must be 0 or 2..22");
    ctx->window_bits = c;
    return FK_OK;
}); }

// ------------------------------------------------------------------------------------------ device buffers
int fk_dev_alloc(fk_ctx *ctx, size_t bytes, void **dptr) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !dptr) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 16));
    return FK_OK;
}); }
int fk_dev_free(fk_ctx *ctx, void *dptr) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (dptr) FK_HIP(ctx, hipFree(dptr));
    return FK_OK;
}); }
int fk_upload(fk_ctx *ctx, void *dptr, const void *host, size_t bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (bytes) FK_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }
int fk_download(fk_ctx *ctx, void *host, const void *dptr, size_t bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (bytes) FK_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }
int fk_dev_copy(fk_ctx *ctx, void *dst, const void *src, size_t bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (bytes) FK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FK_OK;
}); }
// the HIP stream the library queues its main-path work on (quotient, SpMV, fk_dq_*): lets a host that owns other streams
// (RCCL collectives issued by torch.distributed) order against it with events instead of host-side synchronisation
int fk_stream(fk_ctx *ctx, void **out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !out) return FK_ERR_BAD_ARG;
    *out = (void *)ctx->stream;
    return FK_OK;
}); }
int fk_sync(fk_ctx *ctx) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return msm_sync(ctx);
}); }

// ------------------------------------------------------------------------------------------ witness hand-over
// prover.rs:69-76 produces the witness on the host for every proof.  Pinned buffers + two device slots filled on a copy
// stream let the (num_input + num_aux) * 32-byte upload of proof k+1 run underneath proof k.
int fk_host_alloc(fk_ctx *ctx, size_t bytes, void **hptr) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !hptr) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return FK_OK;
}); }
int fk_host_free(fk_ctx *ctx, void *hptr) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (hptr) FK_HIP(ctx, hipHostFree(hptr));
    return FK_OK;
}); }
int fk_witness_upload_async(fk_ctx *ctx, int slot, const void *z_host, size_t bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (slot < 0 || slot > 1 || (bytes && !z_host)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness upload: slot must be 0 or 1, buffer non-null");
    FK_TRY(witness_slot_reserve(ctx, slot, bytes));
    fk_ctx::WitSlot &w = ctx->wslot[slot];
    if (bytes) FK_HIP(ctx, hipMemcpyAsync(w.buf.p, z_host, bytes, hipMemcpyHostToDevice, ctx->copy_st));
    FK_HIP(ctx, hipEventRecord(w.ready, ctx->copy_st));
    return FK_OK;
}); }
// The sharded hand-over (N ranks, one witness): a rank uploads only ITS piece over its PCIe link and the ranks exchange the pieces over
// xGMI (an all-gather: the library's own between the ranks of an fk_multi, RCCL through torch.distributed between processes) -- the
// witness crosses PCIe once instead of N times.
//   fk_witness_slot           room for total_bytes in the slot; its device pointer (valid until a larger witness is handed over) and the
//                             copy stream (hipStream_t) the hand-over is queued on, for a host that issues the collective itself
//   fk_witness_upload_part_async   host bytes [offset, offset + len) of the witness -> the same offsets of the slot, on the copy stream
//   fk_witness_mark_ready     everything queued on the copy stream so far completes the slot: fk_witness_ptr waits for this point
int fk_witness_slot(fk_ctx *ctx, int slot, size_t total_bytes, void **dptr, void **copy_stream) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (slot < 0 || slot > 1) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness slot must be 0 or 1");
    FK_TRY(witness_slot_reserve(ctx, slot, total_bytes));
    if (dptr) *dptr = ctx->wslot[slot].buf.p;
    if (copy_stream) *copy_stream = (void *)ctx->copy_st;
    return FK_OK;
}); }
int fk_witness_upload_part_async(fk_ctx *ctx, int slot, const void *z_host_part, size_t offset, size_t len) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (slot < 0 || slot > 1 || (len && !z_host_part)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness upload: slot must be 0 or 1, buffer non-null");
    fk_ctx::WitSlot &w = ctx->wslot[slot];
    if (!w.ready || offset + len > w.buf.cap) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness upload: piece [%zu, %zu) outside the slot (call fk_witness_slot first)", offset, offset + len);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    if (len) FK_HIP(ctx, hipMemcpyAsync((uint8_t *)w.buf.p + offset, z_host_part, len, hipMemcpyHostToDevice, ctx->copy_st));
    FK_HIP(ctx, hipEventRecord(w.part, ctx->copy_st));
    return FK_OK;
}); }
int fk_witness_mark_ready(fk_ctx *ctx, int slot) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (slot < 0 || slot > 1 || !ctx->wslot[slot].ready) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness slot %d holds nothing", slot);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipEventRecord(ctx->wslot[slot].ready, ctx->copy_st));
    return FK_OK;
}); }
int fk_witness_ptr(fk_ctx *ctx, int slot, void **dptr) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !dptr) return FK_ERR_BAD_ARG;
    if (slot < 0 || slot > 1 || !ctx->wslot[slot].ready) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "witness slot %d holds nothing", slot);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wslot[slot].ready, 0));      // stream-ordered: the host does not wait
    *dptr = ctx->wslot[slot].buf.p;
    return FK_OK;
}); }

}  // extern "C"
namespace fk {
int witness_slot_reserve(fk_ctx *ctx, int slot, size_t bytes, bool *moved) {
    FK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_st) FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_st, hipStreamNonBlocking));
    fk_ctx::WitSlot &w = ctx->wslot[slot];
    if (!w.ready) FK_HIP(ctx, hipEventCreateWithFlags(&w.ready, hipEventDisableTiming));
    if (!w.part) FK_HIP(ctx, hipEventCreateWithFlags(&w.part, hipEventDisableTiming));
    if (moved) *moved = false;
    if (bytes > w.buf.cap) {                      // growing frees the old buffer: nothing may still be reading or filling it
        FK_HIP(ctx, hipStreamSynchronize(ctx->copy_st));
        FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FK_TRY(msm_sync(ctx));
        FK_HIP(ctx, w.buf.reserve(bytes));
        if (moved) *moved = true;
    }
    return FK_OK;
}
// The uploads fk_prove_r1cs_submit deferred: queued on the copy stream, behind the current position of the main stream if
// gate_on_main (the prover calls this when its memory-bound front -- sorts, evaluation of a, b, c -- is queued and the
// VALU-bound part begins).
int upload_deferred(fk_ctx *ctx, bool gate_on_main) {
    for (int s = 0; s < 2; s++) {
        fk_ctx::WitSlot &w = ctx->wslot[s];
        if (!w.deferred) continue;
        w.deferred = false;
        if (gate_on_main) {
            if (!ctx->copy_st) FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_st, hipStreamNonBlocking));
            if (!ctx->ev_upload_gate) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload_gate, hipEventDisableTiming));
            FK_HIP(ctx, hipEventRecord(ctx->ev_upload_gate, ctx->stream));
            FK_HIP(ctx, hipStreamWaitEvent(ctx->copy_st, ctx->ev_upload_gate, 0));
        }
        FK_TRY(fk_witness_upload_async(ctx, s, w.host_z, w.host_bytes));
    }
    return FK_OK;
}
}  // namespace fk
extern "C" {

// ------------------------------------------------------------------------------------------ key
static int key_alloc_slices(fk_ctx *ctx, fk_key *k, double zlo, double zhi) {
    FK_TRY(key_plan_slices(ctx, k, zlo, zhi));
    FK_HIP(ctx, hipMalloc((void **)&k->d_h, (k->h_hi - k->h_lo) * 64 + 64));
    FK_HIP(ctx, hipMalloc((void **)&k->d_l, (k->l_hi - k->l_lo) * 64 + 64));
    FK_HIP(ctx, hipMalloc((void **)&k->d_a, (k->a_hi - k->a_lo) * 64 + 64));
    FK_HIP(ctx, hipMalloc((void **)&k->d_b1, (k->b_hi - k->b_lo) * 64 + 64));
    FK_HIP(ctx, hipMalloc((void **)&k->d_b2, (k->b2_hi - k->b2_lo) * 128 + 128));
    return FK_OK;
}

void fk_key_free(fk_ctx *ctx, fk_key *k) {
    if (!k) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (void *p : {(void *)k->d_h, (void *)k->d_l, (void *)k->d_a, (void *)k->d_b1, (void *)k->d_b2}) if (p) (void)hipFree(p);
    key_pre_free(k);
    delete k;
}

static int key_check_shape(fk_ctx *ctx, uint64_t m, uint32_t num_input, uint64_t n_h, uint64_t n_l, uint32_t num_aux,
                           uint32_t shard_index, uint32_t shard_count) {
    if (m == 0 || (m & (m - 1))) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: m must be a power of two");
    if (m > ((uint64_t)1 << (FK_FR_S - 1))) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "key: m exceeds 2^%d", FK_FR_S - 1);
    if (num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: num_input must include the constant ONE");
    if (n_h != m - 1) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key: h must hold m-1 points");
    if (n_l != num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key: l must hold num_aux points");
    if (shard_count == 0 || shard_index >= shard_count) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: bad shard %u/%u", shard_index, shard_count);
    return FK_OK;
}
int fk_key_load(fk_ctx *ctx, const fk_key_desc *d, fk_key **out) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_key_load");
    if (!ctx || !d || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_TRY(key_check_shape(ctx, d->m, d->num_input, d->n_h, d->n_l, d->num_aux, d->shard_index, d->shard_count));
    if (!d->alpha_g1 || !d->beta_g1 || !d->delta_g1 || !d->beta_g2 || !d->delta_g2) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: missing vk points");
    if ((d->n_h && !d->h) || (d->n_l && !d->l) || (d->n_a && !d->a) || (d->n_b && (!d->b_g1 || !d->b_g2)))
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key: missing key array");
    fk_key *k = new fk_key();
    k->m = d->m; k->num_input = d->num_input; k->num_aux = d->num_aux;
    k->n_h = d->n_h; k->n_l = d->n_l; k->n_a = d->n_a; k->n_b = d->n_b;
    k->shard_index = d->shard_index; k->shard_count = d->shard_count;
    k->alpha_g1 = g1_from_raw(d->alpha_g1); k->beta_g1 = g1_from_raw(d->beta_g1); k->delta_g1 = g1_from_raw(d->delta_g1);
    k->beta_g2 = g2_from_raw(d->beta_g2); k->delta_g2 = g2_from_raw(d->delta_g2);
    int rc = key_alloc_slices(ctx, k, d->z_frac_lo, d->z_frac_hi);
    auto up = [&](void *dst, const uint8_t *src, uint64_t lo, uint64_t hi, size_t w) -> int {
        if (hi > lo) FK_HIP(ctx, hipMemcpy(dst, src + lo * w, (hi - lo) * w, hipMemcpyHostToDevice));
        return FK_OK;
    };
    if (rc == FK_OK) rc = up(k->d_h, d->h, k->h_lo, k->h_hi, 64);
    if (rc == FK_OK) rc = up(k->d_l, d->l, k->l_lo, k->l_hi, 64);
    if (rc == FK_OK) rc = up(k->d_a, d->a, k->a_lo, k->a_hi, 64);
    if (rc == FK_OK) rc = up(k->d_b1, d->b_g1, k->b_lo, k->b_hi, 64);
    if (rc == FK_OK) rc = up(k->d_b2, d->b_g2, k->b2_lo, k->b2_hi, 128);
    if (rc == FK_OK) rc = key_precompute(ctx, k);
    if (rc != FK_OK) { fk_key_free(ctx, k); return rc; }
    *out = k;
    return FK_OK;
}); }

int fk_key_host_vk(const uint8_t *alpha_g1, const uint8_t *beta_g1, const uint8_t *delta_g1, const uint8_t *beta_g2,
                   const uint8_t *delta_g2, fk_key **out) {
    if (!alpha_g1 || !beta_g1 || !delta_g1 || !beta_g2 || !delta_g2 || !out) return FK_ERR_BAD_ARG;
    fk_key *k = new fk_key();
    k->alpha_g1 = g1_from_raw(alpha_g1); k->beta_g1 = g1_from_raw(beta_g1); k->delta_g1 = g1_from_raw(delta_g1);
    k->beta_g2 = g2_from_raw(beta_g2); k->delta_g2 = g2_from_raw(delta_g2);
    *out = k;
    return FK_OK;
}

int fk_key_precomputed(const fk_key *k, uint32_t out[5]) {
    if (!k || !out) return FK_ERR_BAD_ARG;
    const KeyPre *p[5] = {&k->pre_h, &k->pre_l, &k->pre_a, &k->pre_b1, &k->pre_b2};
    for (int i = 0; i < 5; i++) out[i] = p[i]->lev ? p[i]->W : 0;
    return FK_OK;
}

// per array (h, l, a, b_g1, b_g2): levels held, GiB they occupy (or would: negative when the array has none), and an ESTIMATE of the ms
// per proof they save -- 4.6e-8 ms per point of G1 work (a G2 point = FK_G2_WORK), the all-levels-vs-none difference measured at 2^25
// (DESIGN.md section 3.3); 0 for an array too small for levels
int fk_key_levels_plan(const fk_key *k, double out[15]) {
    if (!k || !out) return FK_ERR_BAD_ARG;
    const KeyPre *p[5] = {&k->pre_h, &k->pre_l, &k->pre_a, &k->pre_b1, &k->pre_b2};
    const uint64_t n[5] = {k->h_hi - k->h_lo, k->l_hi - k->l_lo, k->a_hi - k->a_lo, k->b_hi - k->b_lo, k->b2_hi - k->b2_lo};
    for (int i = 0; i < 5; i++) {
        const double pt = i == 4 ? 128.0 : 64.0, work = (i == 4 ? (double)FK_G2_WORK : 1.0) * (double)n[i];
        if (p[i]->lev) { out[3 * i] = p[i]->W; out[3 * i + 1] = (double)(p[i]->W - 1) * (double)n[i] * pt / 1073741824.0; out[3 * i + 2] = work * 4.6e-8; }
        else { out[3 * i] = 0; out[3 * i + 1] = n[i] + n[i] / 2 >= (1ull << 21) ? -11.0 * (double)n[i] * pt / 1073741824.0 : 0.0; out[3 * i + 2] = 0; }
    }
    return FK_OK;
}

// (Re)derives the fixed-base levels of a loaded key against the HBM that is free NOW: for a key loaded with FK_KEY_NO_LEVELS while something else
// was still to be placed in HBM (params_io.load_parameters reads the key while the gate blob is being decoded on the host, uploads the
// constraint system, then calls this), or after memory has been freed.  Not while proofs with this key are in flight.
int fk_key_derive_levels(fk_ctx *ctx, fk_key *k) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !k) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FK_TRY(msm_sync(ctx));
    const auto t0 = std::chrono::steady_clock::now();
    FK_TRY(key_precompute(ctx, k));
    k->load_s[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return FK_OK;
}); }

int fk_key_drop_levels(fk_ctx *ctx, fk_key *k) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !k) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FK_TRY(msm_sync(ctx));
    key_pre_free(k);
    return FK_OK;
}); }

int fk_key_levels_headroom(fk_ctx *ctx, const fk_key *k, int64_t *bytes) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !k || !bytes) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return key_levels_headroom(ctx, k, bytes);
}); }

int fk_key_load_profile(const fk_key *k, double out[2]) {
    if (!k || !out) return FK_ERR_BAD_ARG;
    out[0] = k->load_s[0]; out[1] = k->load_s[1];
    return FK_OK;
}

int fk_key_shard_info(const fk_key *k, uint64_t out[8]) {
    if (!k || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[8] = {k->h_lo, k->h_hi, k->l_lo, k->l_hi, k->a_lo, k->a_hi, k->b_lo, k->b_hi};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

// ... and with b_g2's own slice (it differs from b_g1's in a key split by work): h, l, a, b_g1, b_g2
int fk_key_shard_info2(const fk_key *k, uint64_t out[10]) {
    if (!k || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[10] = {k->h_lo, k->h_hi, k->l_lo, k->l_hi, k->a_lo, k->a_hi, k->b_lo, k->b_hi, k->b2_lo, k->b2_hi};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

int fk_key_synthetic(fk_ctx *ctx, uint64_t m, uint32_t num_input, uint32_t num_aux, uint64_t n_a, uint64_t n_b,
                     uint64_t seed, uint32_t shard_index, uint32_t shard_count, double z_frac_lo, double z_frac_hi, fk_key **out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_TRY(key_check_shape(ctx, m, num_input, m - 1, num_aux, num_aux, shard_index, shard_count));
    fk_key *k = new fk_key();
    k->m = m; k->num_input = num_input; k->num_aux = num_aux;
    k->n_h = m - 1; k->n_l = num_aux; k->n_a = n_a; k->n_b = n_b;
    k->shard_index = shard_index; k->shard_count = shard_count;
    int rc = key_alloc_slices(ctx, k, z_frac_lo, z_frac_hi);
    const uint64_t sd = seed * 1000003ull + shard_index * 7919ull;
    if (rc == FK_OK) rc = gen_points_g1(ctx, k->d_h, k->h_hi - k->h_lo, sd + 1);
    if (rc == FK_OK) rc = gen_points_g1(ctx, k->d_l, k->l_hi - k->l_lo, sd + 2);
    if (rc == FK_OK) rc = gen_points_g1(ctx, k->d_a, k->a_hi - k->a_lo, sd + 3);
    if (rc == FK_OK) rc = gen_points_g1(ctx, k->d_b1, k->b_hi - k->b_lo, sd + 4);
    if (rc == FK_OK) rc = gen_points_g2(ctx, k->d_b2, k->b2_hi - k->b2_lo, sd + 5);
    // vk points: five of the generated points (same on every shard: taken from a seed-only tiny run)
    if (rc == FK_OK) {
        FK_HIP(ctx, ctx->misc.reserve(8 * 128));
        G1Affine *t1 = ctx->misc.as<G1Affine>();
        rc = gen_points_g1(ctx, t1, 3, seed + 99);
        G1Affine h1[3];
        if (rc == FK_OK) { if (hipMemcpy(h1, t1, sizeof h1, hipMemcpyDeviceToHost) != hipSuccess) rc = FK_ERR_HIP; }
        G2Affine *t2 = ctx->misc.as<G2Affine>();
        if (rc == FK_OK) rc = gen_points_g2(ctx, t2, 2, seed + 98);
        G2Affine h2[2];
        if (rc == FK_OK) { if (hipMemcpy(h2, t2, sizeof h2, hipMemcpyDeviceToHost) != hipSuccess) rc = FK_ERR_HIP; }
        k->alpha_g1 = h1[0]; k->beta_g1 = h1[1]; k->delta_g1 = h1[2]; k->beta_g2 = h2[0]; k->delta_g2 = h2[1];
    }
    if (rc == FK_OK) { if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = FK_ERR_HIP; }
    if (rc == FK_OK) rc = key_precompute(ctx, k);
    if (rc != FK_OK) { fk_key_free(ctx, k); return rc; }
    *out = k;
    return FK_OK;
}); }

// ------------------------------------------------------------------------------------------ prover
// L, A, B1, B2 over this key's slices depend on the assignment only, not on the quotient: `witness_begin` queues them on
// the MSM lanes (scalar compaction on an auxiliary stream) WITHOUT touching the main stream, so a quotient that is
// running there -- or its all-to-all phases on several GPUs -- overlaps with their memory-bound sorts and their
// latency-bound tails.  B1 and B2 come first so that the long G2 tail is covered by L and A.  `witness_end` collects
// them (and an H multiplication the caller has begun, if any) into the FK_MSM_RESULT_BYTES record.
// The sorts-first schedule of prove_msms_dev: on for domains of 2^25 and more (below that the phases are short and the
// queue-everything schedule with three lanes is ahead: 2^24 69.6 vs 73.0 ms, 64 transactions 18.2 vs 19.5, 256 transactions
// equal; profiles/r02_sorts_first_probe.log).  FK_PROVE_SORTS_FIRST=0 / 1 forces it off / on.
static bool sorts_first(const fk_key *key) {
    static int t = -2;
    if (t == -2) { const char *e = getenv("FK_PROVE_SORTS_FIRST"); t = e ? atoi(e) : -1; }
    return t >= 0 ? t != 0 : key->h_hi - key->h_lo >= (1ull << 25) - 1;       // by this context's share of H: a rank of a multi-GPU job holds 1/N
}

static int witness_begin(fk_ctx *ctx, const fk_key *key, const Fr *d_z, const uint8_t *d_a_aux, const uint8_t *d_b_in,
                         const uint8_t *d_b_aux, hipEvent_t z_ready) {
    if (!key || !d_z) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (ctx->wit_active) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: witness multiplications already in flight");
    bool any_tail = false;
    for (int i = 0; i < MSM_TAILS; i++) any_tail = any_tail || ctx->tails[i].active;
    if (!any_tail) ctx->lane_next = 0;      // nothing outstanding: start the rotation over (same lanes in every proof)
    // Three lanes keep A from queueing behind B2's long G2 tail: 2^20 14.9 -> 13.3 ms, 2^22 28.0 -> 25.6 ms per proof.  At 2^25
    // it depends on the witness: with mostly trivial B-query scalars (the synthetic shape) the tails are short and a third
    // lane's co-running costs 1 %, with dense ones (1024 rollup transactions: G2 accumulation 35 ms) it gains 2 % -- 216.0 ->
    // 211.2 ms (profiles/r02_lanes_probe.log; one lane, no overlap at all: 227.3).  Since a multiplication is queued without
    // a host round trip (msm.hip) the picture at the top end changed: at 2^25 two lanes are 1 % ahead of three (195.1 vs
    // 197.5 ms, one lane 207.2), below that three still win by 1.5-5 % (profiles/r02_async_msm_ab_probe.log).
    // In the sorts-first schedule (prove_msms_dev) every multiplication has a lane of its own.
    ctx->lanes_in_use = sorts_first(key) ? MSM_LANES : 3;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->aux) {
        FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
    }
    const uint32_t v_in = key->num_input, v_aux = key->num_aux;
    for (int i = 0; i < 4; i++) ctx->wit_tail[i] = -1;
    auto body = [&]() -> int {
        // the gathers of the A / B query scalars: on the auxiliary stream -- or, for an early front (the previous proof's tails are
        // still running and HIP maps more streams than there are hardware queues: the auxiliary stream was seen waiting behind the
        // G2 tail of the B pair's lane), on the main stream, which has nothing queued but the evaluation that follows
        hipStream_t ax = ctx->gather_on_main ? ctx->stream : ctx->aux;
        if (z_ready && !ctx->gather_on_main) FK_HIP(ctx, hipStreamWaitEvent(ax, z_ready, 0));
        // B query: inputs and aux variables that occur in some B-side LC;  A query: all inputs, then the aux variables
        // that occur in some A-side LC
        FK_HIP(ctx, ctx->sc_b.reserve(((size_t)v_in + v_aux) * sizeof(Fr)));
        FK_HIP(ctx, ctx->sc_a.reserve(((size_t)v_in + v_aux) * sizeof(Fr)));
        Fr *sb = ctx->sc_b.as<Fr>(), *sa = ctx->sc_a.as<Fr>();
        const QueryIdx *qi = ctx->qidx;
        if (qi && qi->d_a_aux == d_a_aux && qi->d_b_in == d_b_in && qi->d_b_aux == d_b_aux) {
            // resident constraint system: gather this key's slice of each query straight from the index lists
            if (qi->n_b != key->n_b) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: b query needs %llu points, key holds %llu",
                                                (unsigned long long)qi->n_b, (unsigned long long)key->n_b);
            if (qi->n_a != key->n_a) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: a query needs %llu points, key holds %llu",
                                                (unsigned long long)qi->n_a, (unsigned long long)key->n_a);
            FK_TRY(gather_scalars(ctx, d_z, qi->b + key->b_lo, key->b_hi - key->b_lo, sb + key->b_lo, ax));
            if (key->b2_lo != key->b_lo || key->b2_hi != key->b_hi) {       // key split by work: b_g2's slice is its own (the part b_g1's slice does not cover)
                const uint64_t lo = key->b2_lo, hi = key->b2_hi;
                const uint64_t c_lo = std::max(lo, key->b_lo), c_hi = std::min(hi, key->b_hi);      // overlap with what is gathered already
                if (c_lo >= c_hi) FK_TRY(gather_scalars(ctx, d_z, qi->b + lo, hi - lo, sb + lo, ax));
                else {
                    if (lo < c_lo) FK_TRY(gather_scalars(ctx, d_z, qi->b + lo, c_lo - lo, sb + lo, ax));
                    if (c_hi < hi) FK_TRY(gather_scalars(ctx, d_z, qi->b + c_hi, hi - c_hi, sb + c_hi, ax));
                }
            }
            FK_TRY(gather_scalars(ctx, d_z, qi->a + key->a_lo, key->a_hi - key->a_lo, sa + key->a_lo, ax));
        } else {
            uint64_t n_b_in = 0, n_b_aux = 0, n_a_aux = 0;
            FK_TRY(compact_scalars(ctx, d_z, d_b_in, v_in, sb, &n_b_in, ax));
            FK_TRY(compact_scalars(ctx, d_z + v_in, d_b_aux, v_aux, sb + n_b_in, &n_b_aux, ax));
            if (n_b_in + n_b_aux != key->n_b) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: b query needs %llu points, key holds %llu",
                                                         (unsigned long long)(n_b_in + n_b_aux), (unsigned long long)key->n_b);
            FK_HIP(ctx, hipMemcpyAsync(sa, d_z, (size_t)v_in * sizeof(Fr), hipMemcpyDeviceToDevice, ax));
            FK_TRY(compact_scalars(ctx, d_z + v_in, d_a_aux, v_aux, sa + v_in, &n_a_aux, ax));
            if (v_in + n_a_aux != key->n_a) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: a query needs %llu points, key holds %llu",
                                                       (unsigned long long)(v_in + n_a_aux), (unsigned long long)key->n_a);
        }
        FK_HIP(ctx, hipEventRecord(ctx->ev_aux, ax));
        FK_TRY(msm_g1_begin(ctx, key->d_b1, sb + key->b_lo, key->b_hi - key->b_lo, &ctx->wit_tail[0], ctx->ev_aux, &key->pre_b1));
        // B2: the same scalars as B1 when the two slices coincide (B1's sort is reused); its own slice when the key is split by work
        FK_TRY(msm_g2_begin(ctx, key->d_b2, sb + key->b2_lo, key->b2_hi - key->b2_lo, /*reuse_sort=*/key->b2_lo == key->b_lo && key->b2_hi == key->b_hi,
                            &ctx->wit_tail[1], ctx->ev_aux, &key->pre_b2));
        FK_TRY(msm_g1_begin(ctx, key->d_l, d_z + v_in + key->l_lo, key->l_hi - key->l_lo, &ctx->wit_tail[2], z_ready ? z_ready : ctx->ev_aux, &key->pre_l));   // L reads z itself: it need not wait for the gathers
        FK_TRY(msm_g1_begin(ctx, key->d_a, sa + key->a_lo, key->a_hi - key->a_lo, &ctx->wit_tail[3], ctx->ev_aux, &key->pre_a));
        return FK_OK;
    };
    const int rc = body();
    if (rc != FK_OK) { msm_abandon(ctx); return rc; }
    ctx->wit_active = true;
    return FK_OK;
}

extern "C++" {
namespace fk {
// The witness half of an early front (spmv.hip: early_front): L, A, B1, B2 of the NEXT proof begun with their accumulations and
// tails deferred, exactly as prove_msms_dev's sorts-first path would begin them.  Returns 0 when the schedule does not apply.
int early_witness_begin(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux, const void *d_b_in, const void *d_b_aux, int tails_out[4]) {
    if (!early_front_applies(key)) return 0;
    ctx->defer_back = true;
    ctx->gather_on_main = true;
    const int rc = witness_begin(ctx, key, (const Fr *)d_z, (const uint8_t *)d_a_aux, (const uint8_t *)d_b_in, (const uint8_t *)d_b_aux, ctx->ev_z);
    ctx->gather_on_main = false;
    ctx->defer_back = false;
    if (rc != FK_OK) return -rc;
    memcpy(tails_out, ctx->wit_tail, 4 * sizeof(int));
    return 1;
}
bool early_front_applies(const fk_key *key) {
    static const int t_wfirst = tune("FK_PROVE_WITNESS_FIRST", 1), t_accgate = tune("FK_PROVE_ACC_AFTER_NTT", 1), t_early = tune("FK_PROVE_EARLY_FRONT", 1);
    return t_early && t_wfirst && t_accgate && key && sorts_first(key);
}
}  // namespace fk
}  // extern "C++"

// tails: the four witness multiplications to collect (B1, B2, L, A) when they are no longer the context's current ones -- the
// next proof's may have been begun already (early front); default: the context's
static int witness_end(fk_ctx *ctx, uint8_t out[FK_MSM_RESULT_BYTES], int tail_h = -1, const int *tails = nullptr) {
    if (!tails) {
        if (!ctx->wit_active) { msm_abandon(ctx); FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: no witness multiplications in flight"); }
        ctx->wit_active = false;
        tails = ctx->wit_tail;
    }
    const int tw[4] = {tails[0], tails[1], tails[2], tails[3]};
    G1Xyzz H = G1Xyzz::inf(), L, A, B1; G2Xyzz B2;
    auto body = [&]() -> int {
        FK_TRY(msm_g1_end(ctx, tw[0], &B1));
        FK_TRY(msm_g2_end(ctx, tw[1], &B2));
        FK_TRY(msm_g1_end(ctx, tw[2], &L));
        FK_TRY(msm_g1_end(ctx, tw[3], &A));
        if (tail_h >= 0) FK_TRY(msm_g1_end(ctx, tail_h, &H));
        return FK_OK;
    };
    const int rc = body();
    if (rc != FK_OK) { msm_abandon(ctx); return rc; }
    g1_to_raw(out, H); g1_to_raw(out + 64, L); g1_to_raw(out + 128, A); g1_to_raw(out + 192, B1); g2_to_raw(out + 256, B2);
    return FK_OK;
}

static int prove_msms_z(fk_ctx *ctx, const fk_key *key, const Fr *d_z, const uint8_t *d_a_aux, const uint8_t *d_b_in,
                        const uint8_t *d_b_aux, uint8_t out[FK_MSM_RESULT_BYTES], fk_timings *tm) {
    if (!out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    const double t0 = now_ms();
    FK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->ev_main) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    FK_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));       // whatever produced z on the main stream
    FK_TRY(witness_begin(ctx, key, d_z, d_a_aux, d_b_in, d_b_aux, ctx->ev_main));
    FK_TRY(witness_end(ctx, out));
    if (tm) tm->msm_l_ms = now_ms() - t0;      // the four run interleaved; the sum is reported in the L slot
    return FK_OK;
}

static int prove_msms_dev(fk_ctx *ctx, const fk_key *key, Fr *d_a, Fr *d_b, Fr *d_c, uint64_t n, const Fr *d_z,
                          const uint8_t *d_a_aux, const uint8_t *d_b_in, const uint8_t *d_b_aux, uint8_t out[FK_MSM_RESULT_BYTES],
                          fk_timings *tm) {
    if (!key || !d_a || !d_b || !d_c || !d_z || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (n == 0 || n > key->m || (key->m > 1 && n <= key->m / 2)) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: %llu rows do not match key domain %llu",
                                                      (unsigned long long)n, (unsigned long long)key->m);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_RANGE("prove: queue front, quotient, accumulations; wait for the results");
    const double t0 = now_ms();
    FK_HIP(ctx, ctx->hbuf.reserve(key->m * sizeof(Fr)));
    Fr *d_h = ctx->hbuf.as<Fr>();
    uint64_t m = 0;
    // early front: this proof's witness multiplications were begun (and a, b, c evaluated) while the previous proof's tails ran
    const bool early = ctx->early.done && ctx->early.key == key && ctx->early.d_z == (const void *)d_z;
    if (ctx->early.done && !early) { msm_abandon(ctx); FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: an early front of another proof is outstanding"); }
    ctx->early.done = false;
    if (!early) ctx->lane_next = 0;     // same lane for the same multiplication in every proof: lane buffers keep their sizes
    ctx->lanes_in_use = sorts_first(key) ? MSM_LANES : 3;      // measured, see witness_begin
    // The witness multiplications depend on z only: they are begun right behind the QUEUED quotient, so that their sorts (and what
    // fits of their accumulations) fill the transforms' gaps: 214.5 -> 206.9 ms per proof on the 1024-transaction system
    // (profiles/r02_cusplit_witness_first_probe.log), and at every smaller size measured -- synthetic 2^20 13.4 -> 11.1 ms,
    // 2^22 26.2 -> 23.7, 2^24 80.4 -> 78.0; 64 transactions 21.0 -> 19.1 (profiles/r02_witness_first_by_size.log).  (Round 1
    // measured an earlier form of this overlap as a loss below 2^24.)  FK_PROVE_WITNESS_FIRST=0 puts the quotient first again.
    static const int t_wfirst = tune("FK_PROVE_WITNESS_FIRST", 1);
    const bool wfirst = t_wfirst != 0;
    if (!ctx->ev_main) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    if (!ctx->ev_z) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
    // z is complete here -- or, for a resident constraint system, it already was before a, b, c were evaluated from it
    // (fk_prove_r1cs_dev): the witness multiplications then start together with that evaluation instead of behind it
    if (!ctx->ev_z_recorded) FK_HIP(ctx, hipEventRecord(ctx->ev_z, ctx->stream));
    ctx->ev_z_recorded = false;
    // Sorts first (the kernel timeline of a proof, tools/trace_gantt.py, is what this follows): a sort's workgroups (1024
    // lanes, > 100 KB of LDS) make no headway underneath the transforms (two 64 KB workgroups per compute unit) nor underneath
    // an accumulation (the whole register file) -- a 1.5 ms scatter pass took 30 ms there and the accumulations behind it
    // started when the quotient was done.  So the witness multiplications are queued FIRST, each on a lane of its own, their
    // sorts run beside the evaluation of a, b, c (latency-bound gathers), and the quotient's first kernel waits for the
    // sorts: from then on transforms and accumulations -- all VALU-bound -- share the GPU without anything crawling, and H
    // (its own lane) sorts as soon as the quotient is done.
    // (An experiment of round 2 queued the evaluation of a, b, c BEHIND the sorts and underneath the accumulations: it took 47.6 ms
    // there instead of 12 -- removed.)
    const bool gate = wfirst && sorts_first(key);
    static const int t_accgate = tune("FK_PROVE_ACC_AFTER_NTT", 1);      // the witness accumulations wait for the quotient, see below
    if (early && !gate) { msm_abandon(ctx); FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: early front without the sorts-first schedule"); }
    if (gate) {
        if (early) {
            memcpy(ctx->wit_tail, ctx->early.tails, sizeof ctx->wit_tail);      // begun by early_front (deferred pieces included)
            ctx->wit_active = true;
        } else {
            ctx->defer_back = t_accgate != 0;
            const int rcw = witness_begin(ctx, key, d_z, d_a_aux, d_b_in, d_b_aux, ctx->ev_z);
            ctx->defer_back = false;
            if (rcw != FK_OK) { msm_abandon(ctx); return rcw; }
        }
        for (MsmLane &ln : ctx->lanes)
            if (ln.ev_sorted_valid) { FK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ln.ev_sorted, 0)); ln.ev_sorted_valid = false; }
    }
    { const int rcu = upload_deferred(ctx, true); if (rcu != FK_OK) { if (gate) msm_abandon(ctx); return rcu; } }      // the next proof's witness: underneath what follows
    int rcq;
    { FK_RANGE("prove: queue quotient (6 transforms)"); rcq = quotient_dev(ctx, d_a, d_b, d_c, n, d_h, &m); }          // queued on the main stream, not waited for
    if (rcq != FK_OK) { if (gate) msm_abandon(ctx); return rcq; }
    const double t1 = now_ms();
    if (wfirst) {
        FK_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
        // ... and the transforms make no headway underneath the accumulations either (one pass took 55 ms there): the
        // witness accumulations are queued behind the quotient.  H's sort then has the whole of them to crawl underneath
        // and a clear run beside their latency-bound tails.
        // The accumulations of B1, L and A then run side by side on their lanes (they fill each other's last waves; one after
        // the other on a stream of their own was measured slower: 192.3 - 195.5 against 185.8 - 188.5 ms,
        // profiles/r02_sorts_first_probe.log), B2's right behind B1's, all tails behind the accumulations of their lane.
        // (H's sort right behind the quotient, ALONE, with every accumulation behind it was measured too: slower by 3-10 ms.)
        if (gate) { const int rcd = msm_run_deferred(ctx, ctx->ev_main); if (rcd != FK_OK) { msm_abandon(ctx); return rcd; } }
        if (!gate) {
            const int rcw = witness_begin(ctx, key, d_z, d_a_aux, d_b_in, d_b_aux, ctx->ev_z);
            if (rcw != FK_OK) { msm_abandon(ctx); return rcw; }
        }
        int t_h0 = -1;
        const int rch = msm_g1_begin(ctx, key->d_h, d_h + key->h_lo, key->h_hi - key->h_lo, &t_h0, ctx->ev_main, &key->pre_h);
        if (rch != FK_OK) { msm_abandon(ctx); return rch; }
        const double t2w = now_ms();
        // Everything of this proof is queued.  Before blocking on its results: the front of the NEXT proof, if the caller has one
        // waiting (fk_prove_r1cs_wait) -- its witness multiplications become the context's current ones, so this proof's are
        // collected by their handles.
        if (gate && ctx->before_block) {
            int mine[4];
            memcpy(mine, ctx->wit_tail, sizeof mine);
            ctx->wit_active = false;
            std::function<int()> hook;
            hook.swap(ctx->before_block);
            int rce;
            { FK_RANGE("prove: queue the NEXT proof's early front"); rce = hook(); }
            if (rce != FK_OK) { msm_abandon(ctx); return rce; }
            { FK_RANGE("prove: wait for the five multiplications"); FK_TRY(witness_end(ctx, out, t_h0, mine)); }
        } else { FK_RANGE("prove: wait for the five multiplications"); FK_TRY(witness_end(ctx, out, t_h0)); }
        if (tm) { tm->ntt_ms = t1 - t0; tm->msm_l_ms = t2w - t1; tm->msm_h_ms = now_ms() - t2w; tm->total_ms = now_ms() - t0; }
        return FK_OK;
    }
    // Single GPU: the multiplications start after the quotient.  Running the witness multiplications underneath it was
    // measured (2^20 .. 2^25): both sides are VALU-bound, nothing is gained at 2^25 and 10 % is lost at 2^20 / 2^22.  So was
    // running only their SORTS underneath it, accumulations held back until it is done: 2^25 139.5 -> 148.1 ms, 2^22 29.2 -> 34.0
    // (the B pair's sort alone: 138.6 -> 152.0 ms) -- an NTT pass moves 2.2 TB/s and holds 64 KB of LDS per workgroup, which the sort's workgroups compete for.
    if (!ctx->ev_main) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    FK_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
    // H first: it is the longest accumulation and hides the sorts of the multiplications behind it (B1, B2, H, L, A -- H after
    // the B pair, so that only B1's smaller sort is exposed -- measured 3 % slower at 2^25 and 13 % slower at 2^22)
    int t_h = -1;
    FK_TRY(msm_g1_begin(ctx, key->d_h, d_h + key->h_lo, key->h_hi - key->h_lo, &t_h, ctx->ev_main, &key->pre_h));
    const int rc = witness_begin(ctx, key, d_z, d_a_aux, d_b_in, d_b_aux, ctx->ev_main);
    if (rc != FK_OK) { msm_abandon(ctx); return rc; }
    const double t2 = now_ms();
    FK_TRY(witness_end(ctx, out, t_h));
    if (tm) { tm->ntt_ms = t1 - t0; tm->msm_l_ms = t2 - t1; tm->msm_h_ms = now_ms() - t2; tm->total_ms = now_ms() - t0; }
    return FK_OK;
}

int fk_prove_msms_z_dev(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux, const void *d_b_in,
                        const void *d_b_aux, uint8_t out[FK_MSM_RESULT_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (tm) memset(tm, 0, sizeof *tm);
    return prove_msms_z(ctx, key, (const Fr *)d_z, (const uint8_t *)d_a_aux, (const uint8_t *)d_b_in, (const uint8_t *)d_b_aux, out, tm);
}); }

int fk_prove_msm_h_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, uint8_t out[FK_G1_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !out || (!d_h_slice && key->h_hi > key->h_lo)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    G1Xyzz H;
    FK_TRY(msm_g1_dev(ctx, key->d_h, (const Fr *)d_h_slice, key->h_hi - key->h_lo, &H, &key->pre_h));
    g1_to_raw(out, H);
    return FK_OK;
}); }

int fk_prove_msm_array_dev(fk_ctx *ctx, const fk_key *key, int which, const void *d_scalars, uint8_t *out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !out || which < FK_ARRAY_H || which > FK_ARRAY_B_G2) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "msm over a key array: null argument or unknown array");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = which == FK_ARRAY_H ? key->h_hi - key->h_lo : which == FK_ARRAY_L ? key->l_hi - key->l_lo
                     : which == FK_ARRAY_A ? key->a_hi - key->a_lo : which == FK_ARRAY_B_G1 ? key->b_hi - key->b_lo : key->b2_hi - key->b2_lo;
    if (!d_scalars && n) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "msm over a key array: null scalars");
    if (which == FK_ARRAY_B_G2) {
        G2Xyzz R;
        FK_TRY(msm_g2_dev(ctx, key->d_b2, (const Fr *)d_scalars, n, &R, false, &key->pre_b2));
        g2_to_raw(out, R);
        return FK_OK;
    }
    const G1Affine *bases = which == FK_ARRAY_H ? key->d_h : which == FK_ARRAY_L ? key->d_l : which == FK_ARRAY_A ? key->d_a : key->d_b1;
    const KeyPre *pre = which == FK_ARRAY_H ? &key->pre_h : which == FK_ARRAY_L ? &key->pre_l : which == FK_ARRAY_A ? &key->pre_a : &key->pre_b1;
    G1Xyzz R;
    FK_TRY(msm_g1_dev(ctx, bases, (const Fr *)d_scalars, n, &R, pre));
    g1_to_raw(out, R);
    return FK_OK;
}); }

int fk_prove_msms_z_begin_dev(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux, const void *d_b_in, const void *d_b_aux) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->ev_main) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    FK_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
    return witness_begin(ctx, key, (const Fr *)d_z, (const uint8_t *)d_a_aux, (const uint8_t *)d_b_in, (const uint8_t *)d_b_aux, ctx->ev_main);
}); }

int fk_prove_msms_finish_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, uint8_t out[FK_MSM_RESULT_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !out || (!d_h_slice && key->h_hi > key->h_lo)) { msm_abandon(ctx); FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument"); }
    FK_HIP(ctx, hipSetDevice(ctx->device));
    int t_h = -1;
    const int rc = msm_g1_begin(ctx, key->d_h, (const Fr *)d_h_slice, key->h_hi - key->h_lo, &t_h, nullptr, &key->pre_h);
    if (rc != FK_OK) { msm_abandon(ctx); return rc; }
    return witness_end(ctx, out, t_h);
}); }

int fk_prove_msms_hz_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, const void *d_z, const void *d_a_aux, const void *d_b_in,
                         const void *d_b_aux, uint8_t out[FK_MSM_RESULT_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (tm) memset(tm, 0, sizeof *tm);
    FK_TRY(fk_prove_msms_z_begin_dev(ctx, key, d_z, d_a_aux, d_b_in, d_b_aux));
    return fk_prove_msms_finish_dev(ctx, key, d_h_slice, out);
}); }

int fk_prove_msms_dev(fk_ctx *ctx, const fk_key *key, void *d_a, void *d_b, void *d_c, uint64_t n, const void *d_z,
                      const void *d_a_aux, const void *d_b_in, const void *d_b_aux, uint8_t out[FK_MSM_RESULT_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (tm) memset(tm, 0, sizeof *tm);
    return prove_msms_dev(ctx, key, (Fr *)d_a, (Fr *)d_b, (Fr *)d_c, n, (const Fr *)d_z, (const uint8_t *)d_a_aux,
                          (const uint8_t *)d_b_in, (const uint8_t *)d_b_aux, out, tm);
}); }

// A = alpha + A_q + r*delta1 ; B = beta2 + B2 + s*delta2 ;
// C = H + L + s*A_q + r*B1 + s*alpha + r*beta1 + (r s)*delta1   
int fk_prove_assemble(fk_ctx *ctx, const fk_key *key, const uint8_t *parts, uint32_t n_parts, const uint64_t r_[4],
                      const uint64_t s_[4], uint8_t out[FK_PROOF_BYTES]) { return fk_guard(ctx, [&]() -> int {
    fk_ctx local;                  // host-only routine: usable without a GPU context
    if (!ctx) ctx = &local;
    if (!key || !parts || !n_parts || !r_ || !s_ || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "assemble: null argument");
    if (key->delta_g1.is_inf() || key->delta_g2.is_inf()) FK_SET_ERR(ctx, FK_ERR_UNEXPECTED_IDENTITY, "delta is the identity");
    G1Xyzz H = G1Xyzz::inf(), L = G1Xyzz::inf(), A = G1Xyzz::inf(), B1 = G1Xyzz::inf();
    G2Xyzz B2 = G2Xyzz::inf();
    for (uint32_t i = 0; i < n_parts; i++) {
        const uint8_t *p = parts + (size_t)i * FK_MSM_RESULT_BYTES;
        H.add_mixed(g1_from_raw(p)); L.add_mixed(g1_from_raw(p + 64)); A.add_mixed(g1_from_raw(p + 128));
        B1.add_mixed(g1_from_raw(p + 192)); B2.add_mixed(g2_from_raw(p + 256));
    }
    Fr rm, sm; memcpy(&rm, r_, 32); memcpy(&sm, s_, 32);
    const Fr rc = Fr::from_mont(rm), sc = Fr::from_mont(sm), rsc = Fr::from_mont(Fr::mul(rm, sm));
    const G1Xyzz d1 = G1Xyzz::from_affine(key->delta_g1), a1 = G1Xyzz::from_affine(key->alpha_g1), b1 = G1Xyzz::from_affine(key->beta_g1);
    const G2Xyzz d2 = G2Xyzz::from_affine(key->delta_g2);
    G1Xyzz gA = G1Xyzz::mul_scalar(d1, rc.v); gA.add_mixed(key->alpha_g1); gA.add(A);
    G2Xyzz gB = G2Xyzz::mul_scalar(d2, sc.v); gB.add_mixed(key->beta_g2); gB.add(B2);
    G1Xyzz gC = G1Xyzz::mul_scalar(d1, rsc.v);
    gC.add(G1Xyzz::mul_scalar(a1, sc.v));
    gC.add(G1Xyzz::mul_scalar(b1, rc.v));
    gC.add(G1Xyzz::mul_scalar(A, sc.v));
    gC.add(G1Xyzz::mul_scalar(B1, rc.v));
    gC.add(H); gC.add(L);
    const G1Affine pa = gA.to_affine(), pc = gC.to_affine();
    const G2Affine pb = gB.to_affine();
    // fawkes Borsh: canonical LE coordinates, infinity = zeros (Fq::from_mont(0) == 0)
    const Fq words[8] = {Fq::from_mont(pa.x), Fq::from_mont(pa.y), Fq::from_mont(pb.x.c0), Fq::from_mont(pb.x.c1),
                         Fq::from_mont(pb.y.c0), Fq::from_mont(pb.y.c1), Fq::from_mont(pc.x), Fq::from_mont(pc.y)};
    memcpy(out, words, 256);
    return FK_OK;
}); }

int fk_prove_dev(fk_ctx *ctx, const fk_key *key, void *d_a, void *d_b, void *d_c, uint64_t n, const void *d_z,
                 const void *d_a_aux, const void *d_b_in, const void *d_b_aux, const uint64_t r[4], const uint64_t s[4],
                 uint8_t out[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (key && key->shard_count != 1) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "fk_prove needs an unsharded key (use fk_prove_msms + fk_prove_assemble)");
    uint8_t msms[FK_MSM_RESULT_BYTES];
    FK_TRY(fk_prove_msms_dev(ctx, key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, msms, tm));
    const double t0 = now_ms();
    FK_TRY(fk_prove_assemble(ctx, key, msms, 1, r, s, out));
    if (tm) { tm->assemble_ms = now_ms() - t0; tm->total_ms += tm->assemble_ms; }
    return FK_OK;
}); }

static int stage_inputs(fk_ctx *ctx, const fk_key *key, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n,
                        const uint64_t *z, const uint8_t *a_aux, const uint8_t *b_in, const uint8_t *b_aux) {
    if (!key || !a || !b || !c || !z || !a_aux || !b_in || !b_aux) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (n == 0 || n > key->m) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: %llu rows do not fit key domain %llu", (unsigned long long)n, (unsigned long long)key->m);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mb = key->m * sizeof(Fr), nb = n * sizeof(Fr);
    const size_t nv = (size_t)key->num_input + key->num_aux;
    FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
    FK_HIP(ctx, ctx->stage_z.reserve(nv * sizeof(Fr)));
    FK_HIP(ctx, ctx->stage_d.reserve(nv + (size_t)key->num_aux + 64));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, a, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_b.p, b, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_c.p, c, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_z.p, z, nv * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    uint8_t *dd = ctx->stage_d.as<uint8_t>();
    if (key->num_aux) FK_HIP(ctx, hipMemcpyAsync(dd, a_aux, key->num_aux, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(dd + key->num_aux, b_in, key->num_input, hipMemcpyHostToDevice, ctx->stream));
    if (key->num_aux) FK_HIP(ctx, hipMemcpyAsync(dd + key->num_aux + key->num_input, b_aux, key->num_aux, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}

int fk_prove_msms(fk_ctx *ctx, const fk_key *key, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n,
                  const uint64_t *z, const uint8_t *a_aux, const uint8_t *b_in, const uint8_t *b_aux,
                  uint8_t out[FK_MSM_RESULT_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    const double t0 = now_ms();
    FK_TRY(stage_inputs(ctx, key, a, b, c, n, z, a_aux, b_in, b_aux));
    const double up = now_ms() - t0;
    uint8_t *dd = ctx->stage_d.as<uint8_t>();
    FK_TRY(fk_prove_msms_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, n, ctx->stage_z.p, dd, dd + key->num_aux,
                             dd + key->num_aux + key->num_input, out, tm));
    if (tm) { tm->upload_ms = up; tm->total_ms += up; }
    return FK_OK;
}); }

int fk_prove(fk_ctx *ctx, const fk_key *key, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n,
             const uint64_t *z, const uint8_t *a_aux, const uint8_t *b_in, const uint8_t *b_aux, const uint64_t r[4],
             const uint64_t s[4], uint8_t out[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (key && key->shard_count != 1) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "fk_prove needs an unsharded key (use fk_prove_msms + fk_prove_assemble)");
    if (!r || !s || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    uint8_t msms[FK_MSM_RESULT_BYTES];
    FK_TRY(fk_prove_msms(ctx, key, a, b, c, n, z, a_aux, b_in, b_aux, msms, tm));
    const double t0 = now_ms();
    FK_TRY(fk_prove_assemble(ctx, key, msms, 1, r, s, out));
    if (tm) { tm->assemble_ms = now_ms() - t0; tm->total_ms += tm->assemble_ms; }
    return FK_OK;
}); }

// ------------------------------------------------------------------------------------------ building blocks
int fk_fr_mul_batch(fk_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (n && (!a || !b || !out)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    if (!n) return FK_OK;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = n * sizeof(Fr);
    FK_HIP(ctx, ctx->stage_a.reserve(bytes)); FK_HIP(ctx, ctx->stage_b.reserve(bytes));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_b.p, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    FK_TRY(fr_mul_batch_dev(ctx, ctx->stage_a.as<Fr>(), ctx->stage_b.as<Fr>(), ctx->stage_a.as<Fr>(), n));
    FK_HIP(ctx, hipMemcpyAsync(out, ctx->stage_a.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }

int fk_ntt_dev(fk_ctx *ctx, void *d_data, uint32_t log_n, int inverse, int coset) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_data) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return ntt_exec_simple(ctx, (Fr *)d_data, log_n, inverse != 0, coset != 0);
}); }

int fk_ntt(fk_ctx *ctx, uint64_t *data, uint32_t log_n, int inverse, int coset) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!data) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    if (log_n >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_n, FK_FR_S - 1);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(Fr) << log_n;
    FK_HIP(ctx, ctx->ntt_io.reserve(bytes));
    FK_HIP(ctx, hipMemcpyAsync(ctx->ntt_io.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    FK_TRY(ntt_exec_simple(ctx, ctx->ntt_io.as<Fr>(), log_n, inverse != 0, coset != 0));
    FK_HIP(ctx, hipMemcpyAsync(data, ctx->ntt_io.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }

int fk_quotient_h_dev(fk_ctx *ctx, void *d_a, void *d_b, void *d_c, uint64_t n, void *d_h_out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_a || !d_b || !d_c || !d_h_out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return quotient_dev(ctx, (Fr *)d_a, (Fr *)d_b, (Fr *)d_c, n, (Fr *)d_h_out, nullptr);
}); }

int fk_dq_gather_dev(fk_ctx *ctx, const void *d_full, uint64_t n, uint32_t log_m, uint32_t rank, uint32_t log_w, void *d_local) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_full || !d_local) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return dq_gather(ctx, (const Fr *)d_full, n, log_m, rank, log_w, (Fr *)d_local);
}); }

int fk_dq_local_dev(fk_ctx *ctx, void *d_x, const void *d_xb, const void *d_xc, uint32_t log_m, uint32_t rank, uint32_t log_w, int stage) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_x) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    if (log_m >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_m, FK_FR_S - 1);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return dq_local(ctx, (Fr *)d_x, (const Fr *)d_xb, (const Fr *)d_xc, log_m, rank, log_w, stage);
}); }

int fk_dq_cross_dev(fk_ctx *ctx, void *d_buf, uint32_t log_m, uint32_t rank, uint32_t log_w, int mode) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_buf) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    if (log_m >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_m, FK_FR_S - 1);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return dq_cross(ctx, (Fr *)d_buf, log_m, rank, log_w, mode);
}); }

int fk_dq_cross_sub_dev(fk_ctx *ctx, void *d_buf, const void *d_sub, uint32_t log_m, uint32_t rank, uint32_t log_w) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!d_buf || !d_sub) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    if (log_m >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_m, FK_FR_S - 1);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    return dq_cross(ctx, (Fr *)d_buf, log_m, rank, log_w, 1, (const Fr *)d_sub);
}); }

int fk_quotient_h(fk_ctx *ctx, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n, uint64_t *h_out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!a || !b || !c || !h_out || !n) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    const uint32_t log_n = ceil_log2_u64(n);
    if (log_n >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "evaluation domain 2^%u too large (max 2^%d)", log_n, FK_FR_S - 1);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mb = sizeof(Fr) << log_n, nb = n * sizeof(Fr);
    FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
    FK_HIP(ctx, ctx->hbuf.reserve(mb));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, a, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_b.p, b, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_c.p, c, nb, hipMemcpyHostToDevice, ctx->stream));
    FK_TRY(quotient_dev(ctx, ctx->stage_a.as<Fr>(), ctx->stage_b.as<Fr>(), ctx->stage_c.as<Fr>(), n, ctx->hbuf.as<Fr>(), nullptr));
    FK_HIP(ctx, hipMemcpyAsync(h_out, ctx->hbuf.p, mb - sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }

int fk_msm_g1_dev(fk_ctx *ctx, const void *d_bases, const void *d_scalars, size_t n, uint8_t out[FK_G1_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!out || (n && (!d_bases || !d_scalars))) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    G1Xyzz r;
    FK_TRY(msm_g1_dev(ctx, (const G1Affine *)d_bases, (const Fr *)d_scalars, n, &r));
    g1_to_raw(out, r);
    return FK_OK;
}); }
int fk_msm_g2_dev(fk_ctx *ctx, const void *d_bases, const void *d_scalars, size_t n, uint8_t out[FK_G2_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!out || (n && (!d_bases || !d_scalars))) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    G2Xyzz r;
    FK_TRY(msm_g2_dev(ctx, (const G2Affine *)d_bases, (const Fr *)d_scalars, n, &r));
    g2_to_raw(out, r);
    return FK_OK;
}); }

static int msm_host(fk_ctx *ctx, const uint8_t *bases, const uint64_t *scalars, size_t n, size_t w, uint8_t *out) {
    if (!out || (n && (!bases || !scalars))) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_HIP(ctx, ctx->stage_a.reserve(n * w + 16)); FK_HIP(ctx, ctx->stage_z.reserve(n * sizeof(Fr) + 16));
    if (n) {
        FK_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, bases, n * w, hipMemcpyHostToDevice, ctx->stream));
        FK_HIP(ctx, hipMemcpyAsync(ctx->stage_z.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    }
    return w == 64 ? fk_msm_g1_dev(ctx, ctx->stage_a.p, ctx->stage_z.p, n, out) : fk_msm_g2_dev(ctx, ctx->stage_a.p, ctx->stage_z.p, n, out);
}
int fk_msm_g1(fk_ctx *ctx, const uint8_t *bases, const uint64_t *scalars, size_t n, uint8_t out[FK_G1_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    return msm_host(ctx, bases, scalars, n, 64, out);
}); }
int fk_msm_g2(fk_ctx *ctx, const uint8_t *bases, const uint64_t *scalars, size_t n, uint8_t out[FK_G2_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    return msm_host(ctx, bases, scalars, n, 128, out);
}); }

int fk_gen_points_g1_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_TRY(gen_points_g1(ctx, (G1Affine *)d_out, n, seed));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }
int fk_gen_points_g2_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_TRY(gen_points_g2(ctx, (G2Affine *)d_out, n, seed));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }
int fk_gen_scalars_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed, int kind) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    FK_TRY(gen_scalars(ctx, (Fr *)d_out, n, seed, kind));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FK_OK;
}); }

// ------------------------------------------------------------------------------------------ synthesis (host)
static inline void eval_lc(Fr *out, const uint64_t *ptr, const uint32_t *col, const uint64_t *val, uint64_t row, const Fr *z,
                           uint32_t num_input, uint8_t *din, uint8_t *daux) {
    Fr acc = Fr::zero();
    const Fr one = Fr::one();
    for (uint64_t k = ptr[row]; k < ptr[row + 1]; k++) {
        const uint32_t v = col[k];
        if (v < num_input) { if (din) din[v] = 1; } else { if (daux) daux[v - num_input] = 1; }
        Fr t = z[v];
        if (val) {                              // NULL: every coefficient of this matrix is ONE
            Fr cf; memcpy(&cf, val + 4 * k, 32);
            if (cf != one) t = Fr::mul(t, cf);
        }
        acc = Fr::add(acc, t);
    }
    *out = acc;
}

int fk_synthesize(fk_ctx *ctx, const fk_r1cs *cs, const uint64_t *z_, uint64_t *a, uint64_t *b, uint64_t *c,
                  uint8_t *a_aux, uint8_t *b_in, uint8_t *b_aux) { return fk_guard(ctx, [&]() -> int {
    fk_ctx local;                  // host-only routine: usable without a GPU context
    if (!ctx) ctx = &local;
    if (!cs || !z_ || !a || !b || !c || !a_aux || !b_in || !b_aux) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null argument");
    const Fr *z = (const Fr *)z_;
    const uint64_t nv = (uint64_t)cs->num_input + cs->num_aux;
    for (const uint64_t *ptr : {cs->a_ptr, cs->b_ptr, cs->c_ptr}) if (!ptr) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "null row pointer");
    const struct { const uint64_t *ptr; const uint32_t *col; } mats[3] = {{cs->a_ptr, cs->a_col}, {cs->b_ptr, cs->b_col}, {cs->c_ptr, cs->c_col}};
    for (const auto &mt : mats)
        for (uint64_t k = mt.ptr[0]; k < mt.ptr[cs->num_gates]; k++)
            if (mt.col[k] >= nv) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: variable index %u out of range", mt.col[k]);
    memset(a_aux, 0, cs->num_aux); memset(b_in, 0, cs->num_input); memset(b_aux, 0, cs->num_aux);
    for (uint64_t g = 0; g < cs->num_gates; g++) {
        eval_lc((Fr *)(a + 4 * g), cs->a_ptr, cs->a_col, cs->a_val, g, z, cs->num_input, nullptr, a_aux);
        eval_lc((Fr *)(b + 4 * g), cs->b_ptr, cs->b_col, cs->b_val, g, z, cs->num_input, b_in, b_aux);
        eval_lc((Fr *)(c + 4 * g), cs->c_ptr, cs->c_col, cs->c_val, g, z, cs->num_input, nullptr, nullptr);
    }
    for (uint32_t i = 0; i < cs->num_input; i++) {   // bellman appends `input_i * 0 = 0` (App. A.1)
        const uint64_t row = cs->num_gates + i;
        memcpy(a + 4 * row, &z[i], 32); memset(b + 4 * row, 0, 32); memset(c + 4 * row, 0, 32);
    }
    return FK_OK;
}); }

void fk_shard_range(uint64_t n, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi) {
    if (!count || !lo || !hi) return;
    slice(n, index, count, lo, hi);
}

void fk_work_shard_ranges(uint64_t n_l, uint64_t n_a, uint64_t n_b, uint32_t index, uint32_t count, uint64_t out[8]) {
    if (!count || index >= count || !out) return;
    work_slices(n_l, n_a, n_b, index, count, out);
}

void fk_work_shard_ranges_q0(uint64_t n_l, uint64_t n_a, uint64_t n_b, uint64_t m, uint32_t index, uint32_t count, uint64_t out[8]) {
    if (!count || index >= count || !out) return;
    work_slices(n_l, n_a, n_b, index, count, out, count > 1 ? (long double)FK_Q0_HANDICAP * (long double)m : 0.0L);
}

void fk_h_shard_range(uint64_t n_h, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi) {
    if (!count || !lo || !hi) return;
    h_slice(n_h, index, count, lo, hi);
}

// ------------------------------------------------------------------------------------------ stats
int fk_stats_reset(fk_ctx *ctx) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FK_TRY(msm_sync(ctx));
    for (auto *v : {&ctx->ev_acc, &ctx->ev_acc2, &ctx->ev_ntt}) {
        for (auto &ep : *v) { ctx->ev_pool.push_back(ep.a); ctx->ev_pool.push_back(ep.b); }
        v->clear();
    }
    ctx->acc_adds[0] = ctx->acc_adds[1] = 0;
    return FK_OK;
}); }
int fk_stats_get(fk_ctx *ctx, int which, double *ms, uint64_t *launches, uint64_t *units) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (which < 0 || which > 6) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "stats: which must be 0 / 3 / 5 (G1 accumulate), 1 / 4 / 6 (G2 accumulate) or 2 (NTT pass)");
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FK_TRY(msm_sync(ctx));
    std::vector<EventPair> &v = (which == 0 || which == 3 || which == 5) ? ctx->ev_acc : ((which == 1 || which == 4 || which == 6) ? ctx->ev_acc2 : ctx->ev_ntt);
    double t = 0; uint64_t u = 0;
    if (which >= 5) {
        // launches of one kernel that run side by side (the sorts-first schedule: B1, L, A) each span the whole phase: the time
        // the KERNEL took is the union of the launches' intervals, not their sum
        std::vector<std::pair<double, double>> iv;
        for (auto &ep : v) {
            float s = 0, e = 0;
            FK_HIP(ctx, hipEventElapsedTime(&s, v[0].a, ep.a)); FK_HIP(ctx, hipEventElapsedTime(&e, v[0].a, ep.b));
            iv.emplace_back((double)s, (double)e); u += ep.units;
        }
        std::sort(iv.begin(), iv.end());
        double cs = 0, ce = 0; bool open = false;
        for (auto &x : iv) {
            if (!open) { cs = x.first; ce = x.second; open = true; }
            else if (x.first <= ce) ce = std::max(ce, x.second);
            else { t += ce - cs; cs = x.first; ce = x.second; }
        }
        if (open) t += ce - cs;
    } else {
        for (auto &ep : v) { float x = 0; FK_HIP(ctx, hipEventElapsedTime(&x, ep.a, ep.b)); t += x; u += ep.units; }
    }
    if (which == 3 || which == 4) u = ctx->acc_adds[which - 3];
    if (ms) *ms = t;
    if (launches) *launches = v.size();
    if (units) *units = u;
    return FK_OK;
}); }

}  // extern "C"
