// Multi-GPU prover behind the C ABI: ONE call proves on N GPUs of one node (include/fawkes_hip.h: fk_init_devices, fk_multi_*).
//
// The reference's entry point is one call -- `prove(params, pub, sec, circuit)`
// (/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:63-90) -- so the sharded form is one call as well: a
// host that binds this library (the Rust shim, shim/src/lib.rs) needs no process group, no launcher and no collective
// library of its own.  One process, one library context and one worker thread per GPU.  What shards:
//   * every key array: rank g keeps 1/N of h, l, a, b_g1, b_g2 (and derives the fixed-base levels of its slices);
//   * the evaluation of a, b, c: rank g evaluates only the rows t = g (mod N) -- the cyclic slice its part of the quotient
//     starts from (fk_r1cs_eval_slice_dev);
//   * the quotient: every m-point transform is cut once between the ranks (ntt.hip, dq_*), one all-to-all per transform, seven
//     per proof; rank g ends with the block of h whose bases it holds;
//   * the five multi-scalar multiplications: rank g sums its slices; the 384-byte partial results are folded on the host.
// The all-to-all is done here, by the library: every rank PULLS its chunks out of its peers' buffers with
// hipMemcpyPeerAsync -- device-to-device DMA over xGMI -- on an exchange stream of its own, ordered against the producers
// and consumers with HIP events (the host threads only tell each other that an event has been recorded).  The exchange of
// polynomial k runs while the rank-local transform of polynomial k + 1 does, like the RCCL form in parallel.py.
// Device ids may repeat (several ranks on one GPU: how a one-GPU box tests this path; copies are then device-local).
//
// FK_MULTI_TRANSPORT=rccl selects the other transport for the same seven exchanges: RCCL (bound with dlopen: librccl.so.1) --
// one communicator per rank from ncclCommInitAll, every exchange a group of ncclSend / ncclRecv on the rank's exchange stream.
// The ordering against the kernels is the same pair of events; the ranks' rendezvous is RCCL's.  Distinct devices only (a
// communicator cannot name a GPU twice); anything that keeps RCCL from starting falls back to the peer copies with a note in
// fk_multi_last_error.  Default: peer copies -- the one transport a one-GPU box can execute with more than one rank.
#include "r1cs.hpp"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <string.h>
#include <stdlib.h>
#include <dlfcn.h>

using namespace fk;

struct fk_multi_key {
    std::vector<fk_key *> shard;
    double split = FK_Z_EQUAL_SPLIT;      // how the arrays were dealt (recorded by the loader, on the calling thread): decides the schedule of every proof with this key
};
struct fk_multi_r1cs { std::vector<fk_r1cs_dev *> rep; };

namespace fk {

static constexpr int MX_EXCHANGES = 7;       // per proof: three ifft halves, two coset halves, icoset half, block-cyclic -> blocks

// librccl, bound at run time like libbrotlidec: the library has no link-time dependency on it
struct RcclApi {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    RcclApi() {
        void *h = nullptr;
        for (const char *nm : {"librccl.so.1", "librccl.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(h, "ncclSend");
        Recv = (decltype(Recv))dlsym(h, "ncclRecv");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllGather;
    }
};
static const RcclApi &rccl_api() { static RcclApi a; return a; }
static constexpr int NCCL_UINT8 = 1;        // ncclUint8 (rccl.h)

struct MultiWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false, done = true;
    int rc = FK_OK;
};

struct MultiRank {
    hipStream_t xs = nullptr;                           // exchange stream: the pulls of this rank
    hipEvent_t ev_ready[MX_EXCHANGES] = {nullptr};      // main stream: the source buffer of exchange e is complete
    hipEvent_t ev_done[MX_EXCHANGES] = {nullptr};       // exchange stream: this rank's chunks of exchange e have arrived
    std::atomic<uint64_t> posted{0};                    // sequence number of the latest ev_ready this rank has recorded
    DevBuf send[3], recv[3];
    uint64_t buf_elems = 0;
    uint8_t part[FK_MSM_RESULT_BYTES];
};

}  // namespace fk

struct fk_multi {
    int n = 0;
    uint32_t log_w = 0;
    bool pow2 = false;
    std::vector<int> dev;
    std::vector<fk_ctx *> ctx;
    std::deque<MultiWorker> workers;
    std::deque<MultiRank> ranks;
    std::string err;
    std::atomic<int> abort{0};
    uint64_t seq = 0;                                   // exchanges begun so far (same on every rank: they run the same schedule)
    bool host_event_wait = false;                       // FK_MULTI_HOST_EVENTS=1: wait for a peer's event on the host instead of in the stream
    bool force_exchange = false;                        // FK_MULTI_FORCE_EXCHANGE=1: run the distributed schedule (exchanges with itself) even with ONE rank -- test aid
    std::vector<void *> comms;                          // FK_MULTI_TRANSPORT=rccl: one RCCL communicator per rank (empty: peer copies)
    std::vector<void *> comms_z;                        // ... and a second set for the witness all-gather, which is issued on the copy streams while a proof's exchanges may be in flight
    std::vector<int32_t> peer;                          // n x n: what fk_init_devices found for every ordered pair of ranks (FK_PEER_*: fk_multi_topology)
    std::vector<int32_t> peer_hip;                      // ... and the HIP error code of a refused hipDeviceEnablePeerAccess (0 otherwise)
    bool whole_witness = false;                         // FK_MULTI_WITNESS=whole: every rank uploads the whole witness over its own PCIe link (rounds 3-4)
    uint64_t z_bytes_uploaded = 0, z_bytes_gathered = 0;   // per hand-over, summed over the ranks: host -> device, device -> device
    // barrier of the rank threads
    std::mutex bmu;
    std::condition_variable bcv;
    int bcount = 0;
    uint64_t bgen = 0;
    // pipelined proofs (submit / wait)
    struct Pending { bool active = false; const fk_multi_key *key = nullptr; const fk_multi_r1cs *r1cs = nullptr; uint64_t r[4], s[4]; } pending[2];
    int ticket_next = 0;
};

namespace fk {

static void worker_main(fk_multi *M, int rank) {
    MultiWorker &w = M->workers[rank];
    (void)hipSetDevice(M->dev[rank]);
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lk(w.mu);
            w.cv.wait(lk, [&] { return w.has_job || w.quit; });
            if (w.quit) return;
            job.swap(w.job);
            w.has_job = false;
        }
        int rc = FK_ERR_HIP;
        try { rc = job(); } catch (const std::bad_alloc &) { rc = FK_ERR_OOM; } catch (...) { rc = FK_ERR_HIP; }
        if (rc != FK_OK) { M->abort.store(1); std::lock_guard<std::mutex> bl(M->bmu); M->bcv.notify_all(); }
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.rc = rc; w.done = true;
        }
        w.cv.notify_all();
    }
}

// runs fn(rank) on every rank's worker thread (serial = one rank after the other: ranks that share a GPU, host-memory-heavy
// loaders) and returns the first failure, with that rank's error text
static int run_all(fk_multi *M, const std::function<int(int)> &fn, bool serial = false) {
    M->abort.store(0);
    // a rank that is never started in this call (serial mode stops at the first failure) must not be read below with the code of
    // an earlier call: only ranks that fail in THIS call carry a code other than FK_OK
    for (int r = 0; r < M->n; r++) { std::lock_guard<std::mutex> lk(M->workers[r].mu); M->workers[r].rc = FK_OK; }
    { std::lock_guard<std::mutex> bl(M->bmu); M->bcount = 0; }
    auto start = [&](int r) {
        MultiWorker &w = M->workers[r];
        std::lock_guard<std::mutex> lk(w.mu);
        w.job = [&fn, r] { return fn(r); };
        w.has_job = true; w.done = false; w.rc = FK_OK;
        w.cv.notify_all();
    };
    auto finish = [&](int r) {
        MultiWorker &w = M->workers[r];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
        return w.rc;
    };
    int rc = FK_OK, bad = -1;
    if (serial) {
        for (int r = 0; r < M->n && rc == FK_OK; r++) { start(r); rc = finish(r); if (rc != FK_OK) bad = r; }
    } else {
        for (int r = 0; r < M->n; r++) start(r);
        for (int r = 0; r < M->n; r++) { const int x = finish(r); if (x != FK_OK && rc == FK_OK) { rc = x; bad = r; } }
    }
    if (rc != FK_OK) {
        // the rank that failed first may be one that only gave up because another one had (abort): prefer a rank with a message
        for (int r = 0; r < M->n; r++) if (M->workers[r].rc != FK_OK && !M->ctx[r]->err.empty() && M->ctx[r]->err.find("aborted") == std::string::npos) { bad = r; rc = M->workers[r].rc; break; }
        char b[64]; snprintf(b, sizeof b, "rank %d (device %d): ", bad, bad >= 0 ? M->dev[bad] : -1);
        M->err = std::string(b) + (bad >= 0 ? M->ctx[bad]->err : std::string());
    }
    return rc;
}

static int barrier(fk_multi *M, fk_ctx *ctx) {
    std::unique_lock<std::mutex> lk(M->bmu);
    const uint64_t gen = M->bgen;
    if (++M->bcount == M->n) { M->bcount = 0; M->bgen++; M->bcv.notify_all(); return FK_OK; }
    M->bcv.wait(lk, [&] { return M->bgen != gen || M->abort.load(); });
    if (M->bgen == gen) { M->bcount--; FK_SET_ERR(ctx, FK_ERR_HIP, "aborted: another rank failed"); }
    return FK_OK;
}

// ---- the all-to-all.  chunk p of this rank's dst <- chunk `rank` of rank p's src (what all_to_all_single does), `bytes` per chunk.
// begin: records "src is ready" on the main stream, then queues the pulls on the exchange stream behind every peer's event;
// end: the main stream waits for the pulls.  bufsel / k name the buffers (0 = send[k], 1 = recv[k]) so that a rank can find
// its peers' addresses.
static int xchg_begin(fk_multi *M, int rank, int e, int dst_sel, int src_sel, int k, size_t chunk_bytes, uint64_t seq) {
    fk_ctx *ctx = M->ctx[rank];
    MultiRank &me = M->ranks[rank];
    FK_HIP(ctx, hipEventRecord(me.ev_ready[e], ctx->stream));
    me.posted.store(seq, std::memory_order_release);
    uint8_t *dst = (uint8_t *)(dst_sel ? me.recv[k].p : me.send[k].p);
    if (!M->comms.empty()) {
        // RCCL: this rank's sends read its own source (ready behind ev_ready), its receives write its own destination (free: its
        // last consumer is earlier on the main stream, hence behind ev_ready too); the rendezvous with the peers is RCCL's
        const RcclApi &nc = rccl_api();
        const uint8_t *src = (const uint8_t *)(src_sel ? me.recv[k].p : me.send[k].p);
        FK_HIP(ctx, hipStreamWaitEvent(me.xs, me.ev_ready[e], 0));
        int rc = nc.GroupStart();
        for (int p = 0; p < M->n && rc == 0; p++) {
            rc = nc.Send(src + (size_t)p * chunk_bytes, chunk_bytes, NCCL_UINT8, p, M->comms[rank], me.xs);
            if (rc == 0) rc = nc.Recv(dst + (size_t)p * chunk_bytes, chunk_bytes, NCCL_UINT8, p, M->comms[rank], me.xs);
        }
        const int rce = nc.GroupEnd();
        if (rc == 0) rc = rce;
        if (rc != 0) FK_SET_ERR(ctx, FK_ERR_HIP, "RCCL all-to-all failed: %s", nc.GetErrorString ? nc.GetErrorString(rc) : "?");
        FK_HIP(ctx, hipEventRecord(me.ev_done[e], me.xs));
        return FK_OK;
    }
    for (int i = 0; i < M->n; i++) {
        const int p = (rank + i) % M->n;             // start with the own chunk, then walk the peers in a rotated order (spreads the links)
        MultiRank &peer = M->ranks[p];
        while (peer.posted.load(std::memory_order_acquire) < seq) {
            if (M->abort.load()) FK_SET_ERR(ctx, FK_ERR_HIP, "aborted: another rank failed");
            std::this_thread::yield();
        }
        if (M->host_event_wait) FK_HIP(ctx, hipEventSynchronize(peer.ev_ready[e]));
        else FK_HIP(ctx, hipStreamWaitEvent(me.xs, peer.ev_ready[e], 0));
        const uint8_t *src = (const uint8_t *)(src_sel ? peer.recv[k].p : peer.send[k].p) + (size_t)rank * chunk_bytes;
        if (M->dev[p] == M->dev[rank]) FK_HIP(ctx, hipMemcpyAsync(dst + (size_t)p * chunk_bytes, src, chunk_bytes, hipMemcpyDeviceToDevice, me.xs));
        else FK_HIP(ctx, hipMemcpyPeerAsync(dst + (size_t)p * chunk_bytes, M->dev[rank], src, M->dev[p], chunk_bytes, me.xs));
    }
    FK_HIP(ctx, hipEventRecord(me.ev_done[e], me.xs));
    return FK_OK;
}
static int xchg_end(fk_multi *M, int rank, int e) {
    fk_ctx *ctx = M->ctx[rank];
    FK_HIP(ctx, hipStreamWaitEvent(ctx->stream, M->ranks[rank].ev_done[e], 0));
    return FK_OK;
}

// h = (A*B - C)/Z over the ranks; the cyclic slices of a, b, c are in send[0..2] (L = m / W elements each).  The order of
// calls is parallel.py's quotient_distributed (six transforms: c is subtracted in coefficient space); returns with this rank's
// block h[rank * L, (rank + 1) * L) queued into send[0].
static int quotient_multi(fk_multi *M, int rank, uint32_t log_m, uint64_t seq0) {
    fk_ctx *ctx = M->ctx[rank];
    MultiRank &me = M->ranks[rank];
    const uint32_t lw = M->log_w;
    const size_t chunk = (sizeof(Fr) << (log_m - lw)) >> lw;
    Fr *send[3], *recv[3];
    for (int k = 0; k < 3; k++) { send[k] = me.send[k].as<Fr>(); recv[k] = me.recv[k].as<Fr>(); }
    for (int k = 0; k < 3; k++) {                           // ifft, first half; its exchange starts at once
        FK_TRY(dq_local(ctx, send[k], nullptr, nullptr, log_m, rank, lw, 0));
        FK_TRY(xchg_begin(M, rank, k, 1, 0, k, chunk, seq0 + k + 1));
    }
    for (int k = 0; k < 2; k++) {                           // a, b: ifft second half, coset shift, coset_fft first half
        FK_TRY(xchg_end(M, rank, k));
        FK_TRY(dq_cross(ctx, recv[k], log_m, rank, lw, 0));
        FK_TRY(xchg_begin(M, rank, 3 + k, 0, 1, k, chunk, seq0 + 4 + k));
    }
    FK_TRY(xchg_end(M, rank, 2));
    FK_TRY(dq_cross(ctx, recv[2], log_m, rank, lw, 2));     // c: ifft second half * 1 / (m Z(g)); block-cyclic coefficients stay in recv[2]
    for (int k = 1; k >= 0; k--) {                          // coset_fft, second half (b first: a is multiplied in place next)
        FK_TRY(xchg_end(M, rank, 3 + k));
        FK_TRY(dq_local(ctx, send[k], nullptr, nullptr, log_m, rank, lw, 1));
    }
    FK_TRY(dq_local(ctx, send[0], send[1], nullptr, log_m, rank, lw, 3));      // a*b, icoset_fft first half
    FK_TRY(xchg_begin(M, rank, 5, 1, 0, 0, chunk, seq0 + 6));
    FK_TRY(xchg_end(M, rank, 5));
    FK_TRY(dq_cross(ctx, recv[0], log_m, rank, lw, 1, recv[2]));               // icoset_fft second half, / Z(g), - c's coefficients
    FK_TRY(xchg_begin(M, rank, 6, 0, 1, 0, chunk, seq0 + 7));                  // block-cyclic -> blocks (the key's h sharding)
    FK_TRY(xchg_end(M, rank, 6));
    return FK_OK;
}

static bool overlap_witness(const fk_multi *M) {
    // the witness multiplications do not need the quotient: begun first, they fill the GPU during the exchanges.  On one GPU
    // the overlap measured neutral to slower; from four ranks on the per-rank transforms are short next to seven exchanges.
    const char *e = getenv("FK_OVERLAP_WITNESS");
    if (e && e[0]) return e[0] != '0';
    return M->n >= 4;
}

// one rank's share of a proof; the witness is in this rank's slot `slot`
static int prove_rank(fk_multi *M, int rank, const fk_multi_key *K, const fk_multi_r1cs *R, int slot, const uint64_t r_[4], const uint64_t s_[4],
                      uint8_t out[FK_PROOF_BYTES], fk_timings *tm, uint64_t seq0) {
    fk_ctx *ctx = M->ctx[rank];
    MultiRank &me = M->ranks[rank];
    const fk_key *key = K->shard[rank];
    const fk_r1cs_dev *rs = R->rep[rank];
    FK_HIP(ctx, hipSetDevice(ctx->device));
    void *d_z = nullptr;
    FK_TRY(fk_witness_ptr(ctx, slot, &d_z));
    if (rs->num_input != key->num_input || rs->num_aux != key->num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: constraint system and key disagree on the variable counts");
    const uint64_t rows = rs->num_gates + rs->num_input;
    if (rows > key->m || (key->m > 1 && rows <= key->m / 2)) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: %llu rows do not match key domain %llu",
                                                                     (unsigned long long)rows, (unsigned long long)key->m);
    if (M->n == 1 && !M->force_exchange) return fk_prove_r1cs_dev(ctx, key, rs, d_z, r_, s_, out, tm);       // nothing to cut: the single-GPU prover
    const uint32_t log_m = ceil_log2_u64(key->m);
    uint8_t *part = me.part;
    // the split the key was loaded with says which schedule it is for (recorded in the key -- not inferred from the shape of its slices,
    // ADVICE r4): FK_Z_WORK_SPLIT_Q0 = all of h on rank 0, "quotient on rank 0" (no exchange at all)
    const bool q0 = M->n > 1 && K->split == FK_Z_WORK_SPLIT_Q0;
    if (q0) {
        if (rank == 0) {
            const size_t mb = key->m * sizeof(Fr);
            FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
            FK_TRY(fk_r1cs_eval_dev(ctx, rs, d_z, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p));
            ctx->qidx = &rs->qidx;
            const int rc = fk_prove_msms_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, rows, d_z, rs->d_a_aux, rs->d_b_in, rs->d_b_aux, part, nullptr);
            ctx->qidx = nullptr;
            FK_TRY(rc);
        } else {
            FK_TRY(fk_prove_msms_hz_r1cs_dev(ctx, key, rs, nullptr, d_z, part));          // witness multiplications only: this rank holds no h
        }
    } else if (M->pow2 && log_m >= 2 * M->log_w) {
        const uint64_t L = key->m >> M->log_w;
        if (me.buf_elems < L) {
            FK_HIP(ctx, hipStreamSynchronize(ctx->stream)); FK_HIP(ctx, hipStreamSynchronize(me.xs));
            FK_TRY(barrier(M, ctx));              // nobody may still be pulling from the buffers that are about to move
            for (int k = 0; k < 3; k++) { FK_HIP(ctx, me.send[k].reserve(L * sizeof(Fr))); FK_HIP(ctx, me.recv[k].reserve(L * sizeof(Fr))); }
            // the new size counts only once EVERY rank has its buffers: a rank whose reserve() failed leaves before this
            // barrier, the others wake up in it with the abort -- nobody has recorded L, so the next call finds all ranks in this
            // branch together again (a rank that had already noted L would skip it and wait for an exchange that never comes)
            FK_TRY(barrier(M, ctx));
            me.buf_elems = L;
        }
        const bool ov = overlap_witness(M);
        if (ov) FK_TRY(fk_prove_msms_z_begin_r1cs_dev(ctx, key, rs, d_z));
        int rc = fk_r1cs_eval_slice_dev(ctx, rs, d_z, log_m, rank, M->log_w, me.send[0].p, me.send[1].p, me.send[2].p);
        if (rc == FK_OK) rc = quotient_multi(M, rank, log_m, seq0);
        if (rc != FK_OK) { if (ov) msm_abandon(ctx); return rc; }
        if (ov) FK_TRY(fk_prove_msms_finish_dev(ctx, key, me.send[0].p, part));
        else FK_TRY(fk_prove_msms_hz_r1cs_dev(ctx, key, rs, me.send[0].p, d_z, part));
    } else {
        // a rank count that is not a power of two (or a domain too small to cut): every rank computes the whole quotient and
        // its slices of the five multiplications -- no exchange before the fold
        const size_t mb = key->m * sizeof(Fr);
        FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
        FK_TRY(fk_r1cs_eval_dev(ctx, rs, d_z, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p));
        ctx->qidx = &rs->qidx;
        const int rc = fk_prove_msms_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, rows, d_z, rs->d_a_aux, rs->d_b_in, rs->d_b_aux, part, nullptr);
        ctx->qidx = nullptr;
        FK_TRY(rc);
    }
    // "all-reduce of the partial sums": there is no elliptic-curve reduction operator to hand to a collective, so the N 384-byte
    // records meet in host memory (they are host results already: the window sums are folded on the host) and rank 0 folds them
    FK_TRY(barrier(M, ctx));
    if (rank == 0) {
        std::vector<uint8_t> parts((size_t)M->n * FK_MSM_RESULT_BYTES);
        for (int g = 0; g < M->n; g++) memcpy(parts.data() + (size_t)g * FK_MSM_RESULT_BYTES, M->ranks[g].part, FK_MSM_RESULT_BYTES);
        FK_TRY(fk_prove_assemble(ctx, key, parts.data(), (uint32_t)M->n, r_, s_, out));
    }
    (void)tm;
    return FK_OK;
}

// How the witness arrays are dealt to the ranks: by WORK from two ranks on (FK_Z_WORK_SPLIT: one or two large pieces of l | a | b_g1 | b_g2
// per rank instead of 1 / N of each -- round 4); FK_MULTI_SPLIT=equal restores the equal split of rounds 1-3.
// Which SCHEDULE follows from the split a key was loaded with (prove_rank looks at the key's h slices): with 4 or 8 ranks the transforms
// are cut between the ranks (h in blocks of the domain, seven all-to-alls per proof); with 2 ranks -- where every all-to-all would move
// m * 32 / 4 bytes over ONE link, seven times per proof: more than the transforms it saves -- and with rank counts
// that are not a power of two, rank 0 evaluates a, b, c, computes the whole quotient and H, and the other ranks run witness multiplications
// only: NOTHING is exchanged but the 384-byte partial sums (FK_Z_WORK_SPLIT_Q0; FK_MULTI_SPLIT=work forces the cut-transform schedule
// with the work split on 2 ranks, =equal the equal split of rounds 1-3).
static double multi_split(const fk_multi *M) {
    const char *e = getenv("FK_MULTI_SPLIT");
    if (e && !strcmp(e, "equal")) return FK_Z_EQUAL_SPLIT;
    if (M->n < 2) return FK_Z_EQUAL_SPLIT;
    if (e && !strcmp(e, "work")) return FK_Z_WORK_SPLIT;
    return (M->pow2 && M->n >= 4) ? FK_Z_WORK_SPLIT : FK_Z_WORK_SPLIT_Q0;
}

static int multi_check(fk_multi *M, const fk_multi_key *K, const fk_multi_r1cs *R) {
    if (!K || !R || (int)K->shard.size() != M->n || (int)R->rep.size() != M->n) { M->err = "prove: key / constraint system were not loaded through this fk_multi"; return FK_ERR_BAD_ARG; }
    // before anything is read from the caller's witness buffer: its length is taken from the constraint system, so a system that
    // does not belong to the key must be refused HERE, not after (num_input + num_aux) * 32 bytes of `z` have been uploaded
    for (int g = 0; g < M->n; g++)
        if (!K->shard[g] || !R->rep[g] || K->shard[g]->num_input != R->rep[g]->num_input || K->shard[g]->num_aux != R->rep[g]->num_aux) {
            M->err = "prove: constraint system and key disagree on the variable counts"; return FK_ERR_KEY_MISMATCH;
        }
    return FK_OK;
}

}  // namespace fk

extern "C" {

int fk_init_devices(int n_devices, const int *device_ids, fk_multi **out) {
    if (!out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !device_ids) return FK_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FK_ERR_HIP;       // no GPU: fail loudly, no CPU fallback
    for (int i = 0; i < n_devices; i++) if (device_ids[i] < 0 || device_ids[i] >= ndev) return FK_ERR_BAD_ARG;
    fk_multi *M = new fk_multi();
    M->n = n_devices;
    M->dev.assign(device_ids, device_ids + n_devices);
    M->pow2 = (n_devices & (n_devices - 1)) == 0 && n_devices <= 8;
    while ((1 << M->log_w) < n_devices) M->log_w++;
    { const char *e = getenv("FK_MULTI_HOST_EVENTS"); M->host_event_wait = e && e[0] && e[0] != '0'; }
    { const char *e = getenv("FK_MULTI_FORCE_EXCHANGE"); M->force_exchange = e && e[0] && e[0] != '0'; }
    { const char *e = getenv("FK_MULTI_WITNESS"); M->whole_witness = e && !strcmp(e, "whole"); }
    int rc = FK_OK;
    for (int i = 0; i < n_devices && rc == FK_OK; i++) {
        fk_ctx *c = nullptr;
        rc = fk_init(device_ids[i], &c);
        if (rc == FK_OK) M->ctx.push_back(c);
    }
    // ranks that share a device (a GPU named several times: tests and rehearsals) each need their own scratch on it: the key loaders'
    // HBM planning (key_precompute) multiplies what a rank will allocate by the number of its co-tenants
    for (size_t i = 0; i < M->ctx.size(); i++) {
        int same = 0;
        for (int j = 0; j < n_devices; j++) same += device_ids[j] == device_ids[i];
        M->ctx[i]->co_tenants = same;
    }
    // direct access between every pair of distinct devices (xGMI).  "already enabled" is fine; a refusal leaves the copies staged by the runtime --
    // and is RECORDED per ordered pair (fk_multi_topology), with a note in fk_multi_last_error: a first run on a real node must be able to say
    // whether its exchanges went direct or staged (the result used to be dropped)
    M->peer.assign((size_t)n_devices * n_devices, FK_PEER_SELF);
    M->peer_hip.assign((size_t)n_devices * n_devices, 0);
    {
        int staged = 0, refused = 0;
        for (int i = 0; i < n_devices && rc == FK_OK; i++)
            for (int j = 0; j < n_devices; j++) {
                if (device_ids[i] == device_ids[j]) continue;
                int32_t &st = M->peer[(size_t)i * n_devices + j];
                int can = 0;
                const hipError_t ec = hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]);
                if (ec != hipSuccess || !can) { st = FK_PEER_STAGED; M->peer_hip[(size_t)i * n_devices + j] = (int32_t)ec; staged++; (void)hipGetLastError(); continue; }
                (void)hipSetDevice(device_ids[i]);
                const hipError_t ee = hipDeviceEnablePeerAccess(device_ids[j], 0);
                (void)hipGetLastError();
                if (ee == hipSuccess || ee == hipErrorPeerAccessAlreadyEnabled) st = FK_PEER_DIRECT;
                else { st = FK_PEER_REFUSED; M->peer_hip[(size_t)i * n_devices + j] = (int32_t)ee; refused++; }
            }
        if (staged || refused) {
            char b[200];
            snprintf(b, sizeof b, "note: peer access missing for %d ordered device pairs (%d not reachable, %d refused) -- their copies are staged by the runtime (fk_multi_topology)",
                     staged + refused, staged, refused);
            M->err = b;
        }
    }
    for (int i = 0; i < n_devices && rc == FK_OK; i++) {
        M->ranks.emplace_back();
        MultiRank &rk = M->ranks.back();
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreateWithFlags(&rk.xs, hipStreamNonBlocking) != hipSuccess) { rc = FK_ERR_HIP; break; }
        for (int e = 0; e < MX_EXCHANGES; e++)
            if (hipEventCreateWithFlags(&rk.ev_ready[e], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&rk.ev_done[e], hipEventDisableTiming) != hipSuccess) rc = FK_ERR_HIP;
    }
    if (rc != FK_OK) {
        for (fk_ctx *c : M->ctx) fk_free(c);
        delete M;
        return rc;
    }
    {   // FK_MULTI_TRANSPORT=rccl: communicators for the exchanges (distinct devices only); any obstacle leaves the peer copies in place
        const char *e = getenv("FK_MULTI_TRANSPORT");
        if (e && !strcmp(e, "rccl") && M->pow2) {
            bool distinct = true;
            for (int i = 0; i < n_devices; i++) for (int j = i + 1; j < n_devices; j++) distinct = distinct && device_ids[i] != device_ids[j];
            const RcclApi &nc = rccl_api();
            if (!nc.ok) M->err = "note: FK_MULTI_TRANSPORT=rccl but librccl.so.1 could not be bound -- peer copies are used";
            else if (!distinct) M->err = "note: FK_MULTI_TRANSPORT=rccl needs distinct devices -- peer copies are used";
            else {
                M->comms.assign(n_devices, nullptr);
                const int rcn = nc.CommInitAll(M->comms.data(), n_devices, device_ids);
                if (rcn != 0) { M->comms.clear(); M->err = std::string("note: ncclCommInitAll failed (") + (nc.GetErrorString ? nc.GetErrorString(rcn) : "?") + ") -- peer copies are used"; }
                else {
                    M->comms_z.assign(n_devices, nullptr);
                    if (nc.CommInitAll(M->comms_z.data(), n_devices, device_ids) != 0) { M->comms_z.clear(); M->err = "note: second RCCL communicator set failed -- the witness all-gather uses peer copies"; }
                }
            }
        }
    }
    for (int i = 0; i < n_devices; i++) M->workers.emplace_back();
    for (int i = 0; i < n_devices; i++) M->workers[i].th = std::thread(worker_main, M, i);
    *out = M;
    return FK_OK;
}

void fk_multi_free(fk_multi *M) {
    if (!M) return;
    for (auto &w : M->workers) {
        { std::lock_guard<std::mutex> lk(w.mu); w.quit = true; }
        w.cv.notify_all();
        if (w.th.joinable()) w.th.join();
    }
    for (int i = 0; i < M->n; i++) {
        MultiRank &rk = M->ranks[i];
        (void)hipSetDevice(M->dev[i]);
        (void)hipStreamSynchronize(M->ctx[i]->stream);
        if (rk.xs) { (void)hipStreamSynchronize(rk.xs); (void)hipStreamDestroy(rk.xs); }
        for (int e = 0; e < MX_EXCHANGES; e++) { if (rk.ev_ready[e]) (void)hipEventDestroy(rk.ev_ready[e]); if (rk.ev_done[e]) (void)hipEventDestroy(rk.ev_done[e]); }
        for (int k = 0; k < 3; k++) { rk.send[k].release(); rk.recv[k].release(); }
    }
    for (void *c : M->comms) if (c) (void)rccl_api().CommDestroy(c);
    for (void *c : M->comms_z) if (c) (void)rccl_api().CommDestroy(c);
    for (fk_ctx *c : M->ctx) fk_free(c);
    delete M;
}

// which transport the exchanges use: "rccl" or "peer-dma"
const char *fk_multi_transport(const fk_multi *M) { return (M && !M->comms.empty()) ? "rccl" : "peer-dma"; }

// what fk_init_devices found per ordered pair of ranks: out[i * n + j] = FK_PEER_* for copies INTO rank i's device FROM rank j's
int fk_multi_topology(const fk_multi *M, int32_t *out) {
    if (!M || !out) return FK_ERR_BAD_ARG;
    for (size_t k = 0; k < M->peer.size(); k++) out[k] = M->peer[k];
    return FK_OK;
}

// First contact with a node's links, before any key is built: for every ordered pair of ranks (i, j), i != j, rank j fills `bytes` bytes of its
// device with a pattern of the pair ON ITS MAIN STREAM and records an event; rank i's exchange stream waits for that event (in the stream, or on
// the host with FK_MULTI_HOST_EVENTS) and pulls the bytes -- hipMemcpyPeerAsync between distinct devices, exactly what every exchange of a proof
// does -- timed by an event pair on that stream; the bytes are then compared on the host.  One pair at a time (a link's own rate, not the
// node's).  gbps[i * n + j] = GB/s of the pull, status[i * n + j] = FK_OK / FK_ERR_* (0 on the diagonal).  If the in-stream cross-device wait of
// the first distinct-device pair FAILS and the host-side wait works, the context switches itself to host-side waits (what FK_MULTI_HOST_EVENTS=1
// selects) and says so: *host_events_out = 1 and a note in fk_multi_last_error.  Returns FK_OK when every pair passed.
int fk_multi_preflight(fk_multi *M, size_t bytes, double *gbps, int32_t *status, int *host_events_out) { return fk_guard(M, [&]() -> int {
    if (!M || !gbps || !status || bytes < 64 || bytes > ((size_t)1 << 32) || (bytes & 7)) return FK_ERR_BAD_ARG;
    const int n = M->n;
    for (int k = 0; k < n * n; k++) { gbps[k] = 0.0; status[k] = FK_OK; }
    if (host_events_out) *host_events_out = M->host_event_wait ? 1 : 0;
    std::vector<uint64_t> want(bytes / 8), got(bytes / 8);
    std::vector<void *> buf(n, nullptr), dst(n, nullptr);
    int rc = FK_OK;
    auto fail = [&](int i, int j, int code, const char *what, hipError_t e) {
        status[i * n + j] = code;
        char b[256]; snprintf(b, sizeof b, "preflight: pair (into rank %d / device %d, from rank %d / device %d): %s%s%s", i, M->dev[i], j, M->dev[j], what,
                              e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
        M->err = b;
        (void)hipGetLastError();
        rc = code;
    };
    for (int r = 0; r < n; r++) {
        if (hipSetDevice(M->dev[r]) != hipSuccess || hipMalloc(&buf[r], bytes) != hipSuccess || hipMalloc(&dst[r], bytes) != hipSuccess) { M->err = "preflight: hipMalloc failed"; rc = FK_ERR_OOM; break; }
    }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    for (int i = 0; i < n && rc != FK_ERR_OOM; i++) {
        for (int j = 0; j < n; j++) {
            if (i == j) continue;
            fk_ctx *ci = M->ctx[i], *cj = M->ctx[j];
            MultiRank &me = M->ranks[i], &peer = M->ranks[j];
            uint64_t x = 0x9e3779b97f4a7c15ull * (uint64_t)(i * 64 + j + 1);
            for (auto &w : want) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
            // producer side: rank j's main stream
            hipError_t e = hipSetDevice(M->dev[j]);
            if (e == hipSuccess) e = hipMemcpyAsync(buf[j], want.data(), bytes, hipMemcpyHostToDevice, cj->stream);
            if (e == hipSuccess) e = hipEventRecord(peer.ev_ready[0], cj->stream);
            if (e != hipSuccess) { fail(i, j, FK_ERR_HIP, "producer side", e); continue; }
            // consumer side: rank i's exchange stream
            e = hipSetDevice(M->dev[i]);
            if (e == hipSuccess && !t0) { e = hipEventCreate(&t0); if (e == hipSuccess) e = hipEventCreate(&t1); }
            if (e == hipSuccess) e = hipMemsetAsync(dst[i], 0, bytes, me.xs);
            if (e == hipSuccess) {
                if (M->host_event_wait) e = hipEventSynchronize(peer.ev_ready[0]);
                else {
                    e = hipStreamWaitEvent(me.xs, peer.ev_ready[0], 0);
                    if (e != hipSuccess && M->dev[i] != M->dev[j]) {
                        // the cross-device wait is refused: fall back to the host-side wait for this and every later exchange
                        (void)hipGetLastError();
                        const hipError_t e2 = hipEventSynchronize(peer.ev_ready[0]);
                        if (e2 == hipSuccess) {
                            M->host_event_wait = true;
                            if (host_events_out) *host_events_out = 1;
                            M->err = std::string("note: preflight: hipStreamWaitEvent on another device's event failed (") + hipGetErrorString(e) + ") -- switched to host-side event waits (FK_MULTI_HOST_EVENTS)";
                            e = hipSuccess;
                        }
                    }
                }
            }
            if (e == hipSuccess) e = hipEventRecord(t0, me.xs);
            if (e == hipSuccess) e = M->dev[i] == M->dev[j] ? hipMemcpyAsync(dst[i], buf[j], bytes, hipMemcpyDeviceToDevice, me.xs)
                                                            : hipMemcpyPeerAsync(dst[i], M->dev[i], buf[j], M->dev[j], bytes, me.xs);
            if (e == hipSuccess) e = hipEventRecord(t1, me.xs);
            if (e == hipSuccess) e = hipStreamSynchronize(me.xs);
            if (e != hipSuccess) { fail(i, j, FK_ERR_HIP, "pull", e); continue; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, t0, t1);
            e = hipMemcpy(got.data(), dst[i], bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { fail(i, j, FK_ERR_HIP, "read-back", e); continue; }
            if (memcmp(got.data(), want.data(), bytes) != 0) { fail(i, j, FK_ERR_HIP, "the pulled bytes differ from the bytes written", hipSuccess); continue; }
            gbps[i * n + j] = ms > 0.f ? (double)bytes / (ms * 1e-3) / 1e9 : 0.0;
            (void)ci;
        }
    }
    for (int r = 0; r < n; r++) {
        (void)hipSetDevice(M->dev[r]);
        if (buf[r]) (void)hipFree(buf[r]);
        if (dst[r]) (void)hipFree(dst[r]);
    }
    if (t0) { (void)hipSetDevice(M->dev[0]); (void)hipEventDestroy(t0); (void)hipEventDestroy(t1); }
    return rc;
}); }

const char *fk_multi_last_error(const fk_multi *M) { return M ? M->err.c_str() : "null multi-GPU context"; }
int fk_multi_size(const fk_multi *M) { return M ? M->n : 0; }
fk_ctx *fk_multi_ctx(fk_multi *M, int rank) { return (M && rank >= 0 && rank < M->n) ? M->ctx[rank] : nullptr; }

static bool shares_device(const fk_multi *M) {
    for (int i = 0; i < M->n; i++) for (int j = i + 1; j < M->n; j++) if (M->dev[i] == M->dev[j]) return true;
    return false;
}

// ---------------------------------------------------------------- keys: shard g of N on rank g
int fk_multi_key_load(fk_multi *M, const fk_key_desc *desc, fk_multi_key **out) { return fk_guard(M, [&]() -> int {
    if (!M || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    if (!desc) { M->err = "key: null descriptor"; return FK_ERR_BAD_ARG; }
    fk_multi_key *K = new fk_multi_key();
    K->shard.assign(M->n, nullptr);
    const double split = K->split = multi_split(M);          // (read here, on the calling thread: the workers do not touch the environment)
    const int rc = run_all(M, [&](int r) {
        fk_key_desc d = *desc;
        d.shard_index = (uint32_t)r; d.shard_count = (uint32_t)M->n; d.z_frac_lo = split; d.z_frac_hi = 0;
        return fk_key_load(M->ctx[r], &d, &K->shard[r]);
    }, shares_device(M));
    if (rc != FK_OK) { fk_multi_key_free(M, K); return rc; }
    *out = K;
    return FK_OK;
}); }

int fk_multi_key_load_bellman(fk_multi *M, const uint8_t *buf, size_t len, uint32_t flags, fk_multi_key **out, uint8_t *gamma_g2_out,
                              uint8_t *ic_out, uint32_t ic_cap, uint32_t *n_ic) { return fk_guard(M, [&]() -> int {
    if (!M || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    fk_multi_key *K = new fk_multi_key();
    K->shard.assign(M->n, nullptr);
    const double split = K->split = multi_split(M);
    const int rc = run_all(M, [&](int r) {
        uint32_t nic = 0;
        const int x = fk_key_load_bellman(M->ctx[r], buf, len, flags, (uint32_t)r, (uint32_t)M->n, split, 0, &K->shard[r],
                                          r == 0 ? gamma_g2_out : nullptr, r == 0 ? ic_out : nullptr, r == 0 ? ic_cap : 0, &nic);
        if (r == 0 && n_ic) *n_ic = nic;
        return x;
    }, shares_device(M));
    if (rc != FK_OK) { fk_multi_key_free(M, K); return rc; }
    *out = K;
    return FK_OK;
}); }

static int multi_setup(fk_multi *M, const fk_r1cs *cs, uint32_t copies, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4],
                       const uint64_t gamma[4], const uint64_t delta[4], fk_multi_key **out, uint8_t vk_out[6 * 128], uint8_t *ic_out) {
    if (!M || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    if (!cs || !vk_out || !ic_out) { M->err = "setup: null argument"; return FK_ERR_BAD_ARG; }
    fk_multi_key *K = new fk_multi_key();
    K->shard.assign(M->n, nullptr);
    const size_t n_ic = copies ? 1 + (size_t)copies * (cs->num_input - 1) : cs->num_input;
    const double split = K->split = multi_split(M);
    const int rc = run_all(M, [&](int r) {
        // every rank derives ONLY its shard of the five arrays (setup.hip); the verifying key comes out of each derivation, rank 0's is returned
        std::vector<uint8_t> vk_tmp, ic_tmp;
        uint8_t *vk = vk_out, *ic = ic_out;
        if (r != 0) { vk_tmp.resize(6 * 128); ic_tmp.resize(n_ic * 64 + 64); vk = vk_tmp.data(); ic = ic_tmp.data(); }
        if (copies) return fk_setup_tiled(M->ctx[r], cs, copies, tau, alpha, beta, gamma, delta, (uint32_t)r, (uint32_t)M->n, split, 0, &K->shard[r], vk, ic);
        return fk_setup(M->ctx[r], cs, tau, alpha, beta, gamma, delta, (uint32_t)r, (uint32_t)M->n, split, 0, &K->shard[r], vk, ic);
    }, shares_device(M));
    if (rc != FK_OK) { fk_multi_key_free(M, K); return rc; }
    *out = K;
    return FK_OK;
}
int fk_multi_setup(fk_multi *M, const fk_r1cs *cs, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4], const uint64_t gamma[4],
                   const uint64_t delta[4], fk_multi_key **out, uint8_t vk_out[6 * 128], uint8_t *ic_out) { return fk_guard(M, [&]() -> int {
    return multi_setup(M, cs, 0, tau, alpha, beta, gamma, delta, out, vk_out, ic_out);
}); }
int fk_multi_setup_tiled(fk_multi *M, const fk_r1cs *instance, uint32_t copies, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4],
                         const uint64_t gamma[4], const uint64_t delta[4], fk_multi_key **out, uint8_t vk_out[6 * 128], uint8_t *ic_out) { return fk_guard(M, [&]() -> int {
    if (M && !copies) { M->err = "setup: copies must be at least 1"; return FK_ERR_BAD_ARG; }
    return multi_setup(M, instance, copies, tau, alpha, beta, gamma, delta, out, vk_out, ic_out);
}); }

void fk_multi_key_free(fk_multi *M, fk_multi_key *K) {
    if (!K) return;
    for (size_t r = 0; r < K->shard.size(); r++) if (K->shard[r]) fk_key_free(M && r < M->ctx.size() ? M->ctx[r] : nullptr, K->shard[r]);
    delete K;
}
const fk_key *fk_multi_key_shard(const fk_multi_key *K, int rank) { return (K && rank >= 0 && rank < (int)K->shard.size()) ? K->shard[rank] : nullptr; }

// ---------------------------------------------------------------- constraint system: one replica per GPU
static int multi_r1cs(fk_multi *M, fk_multi_r1cs **out, const std::function<int(int, fk_r1cs_dev **)> &load) {
    if (!M || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    fk_multi_r1cs *R = new fk_multi_r1cs();
    R->rep.assign(M->n, nullptr);
    const int rc = run_all(M, [&](int r) { return load(r, &R->rep[r]); }, true);      // one after the other: each load builds host-side tables
    if (rc != FK_OK) { fk_multi_r1cs_free(M, R); return rc; }
    *out = R;
    return FK_OK;
}
int fk_multi_r1cs_load(fk_multi *M, const fk_r1cs *cs, fk_multi_r1cs **out) { return fk_guard(M, [&]() -> int {
    return multi_r1cs(M, out, [&](int r, fk_r1cs_dev **o) { return fk_r1cs_load(M->ctx[r], cs, o); });
}); }
int fk_multi_r1cs_load_tiled(fk_multi *M, const fk_r1cs *instance, uint32_t copies, fk_multi_r1cs **out) { return fk_guard(M, [&]() -> int {
    return multi_r1cs(M, out, [&](int r, fk_r1cs_dev **o) { return fk_r1cs_load_tiled(M->ctx[r], instance, copies, o); });
}); }
int fk_multi_r1cs_load_gates(fk_multi *M, const fk_gates *gates, fk_multi_r1cs **out) { return fk_guard(M, [&]() -> int {
    return multi_r1cs(M, out, [&](int r, fk_r1cs_dev **o) { return fk_r1cs_load_gates(M->ctx[r], gates, o); });
}); }
void fk_multi_r1cs_free(fk_multi *M, fk_multi_r1cs *R) {
    if (!R) return;
    for (size_t r = 0; r < R->rep.size(); r++) if (R->rep[r]) fk_r1cs_free(M && r < M->ctx.size() ? M->ctx[r] : nullptr, R->rep[r]);
    delete R;
}
const fk_r1cs_dev *fk_multi_r1cs_replica(const fk_multi_r1cs *R, int rank) { return (R && rank >= 0 && rank < (int)R->rep.size()) ? R->rep[rank] : nullptr; }

// ---------------------------------------------------------------- the prover: witness in (host memory) -> 256-byte proof out
// The witness crosses PCIe ONCE: rank g uploads the piece z[g * C, (g + 1) * C) (C = ceil(variables / N) elements) over its own link into
// its slot, then every rank collects the other N - 1 pieces from its peers over xGMI -- an all-gather of (N - 1) / N of the witness
// into every GPU: peer DMA (hipMemcpyPeerAsync on the rank's copy stream, behind the owner's "piece has arrived" event), or ONE grouped
// ncclAllGather with FK_MULTI_TRANSPORT=rccl.  (Rounds 3-4 had every rank upload all of z: N x 1.07 GB per proof out of one pinned
// buffer -- 230 GB/s of host reads at N = 8 for a 38 ms share of the proof; FK_MULTI_WITNESS=whole keeps that form.)  Every rank needs
// all of z: the evaluation of a, b, c reads it for a cyclic row slice, and the L / A / B slices are gathered from it.
static int multi_upload(fk_multi *M, int slot, const uint64_t *z, size_t bytes) {
    auto fail = [&](int r, int rc) { char b[48]; snprintf(b, sizeof b, "rank %d: ", r); M->err = std::string(b) + M->ctx[r]->err; return rc; };
    if (M->n == 1 || M->whole_witness) {
        for (int r = 0; r < M->n; r++) { const int rc = fk_witness_upload_async(M->ctx[r], slot, z, bytes); if (rc != FK_OK) return fail(r, rc); }
        M->z_bytes_uploaded = bytes * (size_t)M->n; M->z_bytes_gathered = 0;
        return FK_OK;
    }
    const size_t n = (size_t)M->n, elems = bytes / sizeof(Fr), C = (elems + n - 1) / n * sizeof(Fr), padded = C * n;
    // a slot that must grow moves: no peer may still be pulling from it (only ever the first hand-over of a larger system)
    bool grow = false;
    for (int r = 0; r < M->n; r++) grow = grow || M->ctx[r]->wslot[slot].buf.cap < padded;
    if (grow) for (int r = 0; r < M->n; r++) if (M->ctx[r]->copy_st) { (void)hipSetDevice(M->dev[r]); if (hipStreamSynchronize(M->ctx[r]->copy_st) != hipSuccess) { M->ctx[r]->err = "witness upload: synchronize failed"; return fail(r, FK_ERR_HIP); } }
    for (int r = 0; r < M->n; r++) { const int rc = witness_slot_reserve(M->ctx[r], slot, padded); if (rc != FK_OK) return fail(r, rc); }
    auto piece = [&](int r, size_t *off, size_t *len) { *off = std::min(bytes, (size_t)r * C); *len = std::min(bytes - *off, C); };
    for (int r = 0; r < M->n; r++) {
        size_t off, len; piece(r, &off, &len);
        const int rc = fk_witness_upload_part_async(M->ctx[r], slot, (const uint8_t *)z + off, off, len);
        if (rc != FK_OK) return fail(r, rc);
    }
    if (!M->comms_z.empty()) {
        const RcclApi &nc = rccl_api();
        int rc = nc.GroupStart();
        for (int r = 0; r < M->n && rc == 0; r++) {
            uint8_t *buf = (uint8_t *)M->ctx[r]->wslot[slot].buf.p;
            rc = nc.AllGather(buf + (size_t)r * C, buf, C, NCCL_UINT8, M->comms_z[r], M->ctx[r]->copy_st);        // in place: every rank's piece sits at its own offset
        }
        const int rce = nc.GroupEnd();
        if (rc == 0) rc = rce;
        if (rc != 0) { M->err = std::string("RCCL all-gather of the witness failed: ") + (nc.GetErrorString ? nc.GetErrorString(rc) : "?"); return FK_ERR_HIP; }
    } else {
        for (int r = 0; r < M->n; r++) {
            fk_ctx *ctx = M->ctx[r];
            if (hipSetDevice(M->dev[r]) != hipSuccess) { ctx->err = "witness gather: hipSetDevice failed"; return fail(r, FK_ERR_HIP); }
            uint8_t *dst = (uint8_t *)ctx->wslot[slot].buf.p;
            for (int i = 1; i < M->n; i++) {
                const int p = (r + i) % M->n;          // rotated order: at any moment the ranks pull from different peers
                size_t off, len; piece(p, &off, &len);
                if (!len) continue;
                hipError_t e = hipStreamWaitEvent(ctx->copy_st, M->ctx[p]->wslot[slot].part, 0);
                const uint8_t *src = (const uint8_t *)M->ctx[p]->wslot[slot].buf.p + off;
                if (e == hipSuccess) e = M->dev[p] == M->dev[r] ? hipMemcpyAsync(dst + off, src, len, hipMemcpyDeviceToDevice, ctx->copy_st)
                                                              : hipMemcpyPeerAsync(dst + off, M->dev[r], src, M->dev[p], len, ctx->copy_st);
                if (e != hipSuccess) { ctx->err = std::string("witness gather: ") + hipGetErrorString(e); (void)hipGetLastError(); return fail(r, FK_ERR_HIP); }
            }
        }
    }
    for (int r = 0; r < M->n; r++) { const int rc = fk_witness_mark_ready(M->ctx[r], slot); if (rc != FK_OK) return fail(r, rc); }
    M->z_bytes_uploaded = bytes; M->z_bytes_gathered = bytes * (n - 1);
    return FK_OK;
}

static int multi_prove_slot(fk_multi *M, const fk_multi_key *K, const fk_multi_r1cs *R, int slot, const uint64_t r_[4], const uint64_t s_[4],
                            uint8_t out[FK_PROOF_BYTES], fk_timings *tm) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t seq0 = M->seq;
    M->seq += MX_EXCHANGES;
    if (tm) memset(tm, 0, sizeof *tm);
    const int rc = run_all(M, [&](int r) { return prove_rank(M, r, K, R, slot, r_, s_, out, tm, seq0); });
    if (rc != FK_OK) {
        // leave every rank in a state from which the next call can start: nothing queued, no multiplication outstanding
        for (int r = 0; r < M->n; r++) { (void)hipSetDevice(M->dev[r]); msm_abandon(M->ctx[r]); M->ctx[r]->qidx = nullptr; (void)hipStreamSynchronize(M->ctx[r]->stream); (void)hipStreamSynchronize(M->ranks[r].xs); }
        for (auto &rk : M->ranks) rk.posted.store(M->seq);
        return rc;
    }
    if (tm) tm->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FK_OK;
}

int fk_multi_prove_r1cs(fk_multi *M, const fk_multi_key *K, const fk_multi_r1cs *R, const uint64_t *z, const uint64_t r_[4], const uint64_t s_[4],
                        uint8_t out[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(M, [&]() -> int {
    FK_RANGE("fk_multi_prove_r1cs");
    if (!M) return FK_ERR_BAD_ARG;
    if (!z || !r_ || !s_ || !out) { M->err = "prove: null argument"; return FK_ERR_BAD_ARG; }
    if (M->pending[0].active || M->pending[1].active) { M->err = "prove: submitted proofs are outstanding (call fk_multi_prove_r1cs_wait first)"; return FK_ERR_BAD_ARG; }
    FK_TRY(multi_check(M, K, R));
    const size_t zb = ((size_t)R->rep[0]->num_input + R->rep[0]->num_aux) * sizeof(Fr);
    FK_TRY(multi_upload(M, 0, z, zb));
    return multi_prove_slot(M, K, R, 0, r_, s_, out, tm);
}); }

int fk_multi_prove_r1cs_submit(fk_multi *M, const fk_multi_key *K, const fk_multi_r1cs *R, const uint64_t *z, const uint64_t r_[4], const uint64_t s_[4], int *ticket) { return fk_guard(M, [&]() -> int {
    if (!M) return FK_ERR_BAD_ARG;
    if (!z || !r_ || !s_ || !ticket) { M->err = "prove: null argument"; return FK_ERR_BAD_ARG; }
    FK_TRY(multi_check(M, K, R));
    const int slot = M->ticket_next;
    if (M->pending[slot].active) { M->err = "prove: two proofs are already submitted (call fk_multi_prove_r1cs_wait first)"; return FK_ERR_BAD_ARG; }
    const size_t zb = ((size_t)R->rep[0]->num_input + R->rep[0]->num_aux) * sizeof(Fr);
    FK_TRY(multi_upload(M, slot, z, zb));
    fk_multi::Pending &p = M->pending[slot];
    p.active = true; p.key = K; p.r1cs = R; memcpy(p.r, r_, 32); memcpy(p.s, s_, 32);
    M->ticket_next = slot ^ 1;
    *ticket = slot;
    return FK_OK;
}); }

int fk_multi_prove_r1cs_wait(fk_multi *M, int ticket, uint8_t out[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(M, [&]() -> int {
    if (!M) return FK_ERR_BAD_ARG;
    if (ticket < 0 || ticket > 1 || !M->pending[ticket].active || !out) { M->err = "prove: no submitted proof with this ticket"; return FK_ERR_BAD_ARG; }
    fk_multi::Pending &p = M->pending[ticket];
    p.active = false;
    return multi_prove_slot(M, p.key, p.r1cs, ticket, p.r, p.s, out, tm);
}); }

// bytes the latest witness hand-over moved, summed over the ranks: out[0] host -> device (PCIe), out[1] device -> device (the all-gather)
int fk_multi_witness_traffic(const fk_multi *M, uint64_t out[2]) {
    if (!M || !out) return FK_ERR_BAD_ARG;
    out[0] = M->z_bytes_uploaded; out[1] = M->z_bytes_gathered;
    return FK_OK;
}

// waits for everything queued on every rank
int fk_multi_sync(fk_multi *M) { return fk_guard(M, [&]() -> int {
    if (!M) return FK_ERR_BAD_ARG;
    for (int r = 0; r < M->n; r++) {
        const int rc = fk_sync(M->ctx[r]);
        if (rc != FK_OK) { M->err = M->ctx[r]->err; return rc; }
        (void)hipStreamSynchronize(M->ranks[r].xs);
    }
    return FK_OK;
}); }

}  // extern "C"
