// The circuit half of a fawkes `Parameters` file: brotli blob <-> Borsh gate stream <-> CSR with dictionary-coded
// coefficients.  Native, streamed and multi-threaded: the benchmark's 1741-transaction system is 3.36e7 gates / 1.64e9 terms =
// 61 GB of decoded stream behind a 2.8 GB blob -- a per-term interpreter loop, or a host copy of the decoded stream, is not an
// option.
//
// Replaces, at key-load time, what fawkes does again for EVERY proof:
//   WitnessCS::get_gate_iterator   /root/reference/fawkes-crypto/src/circuit/r1cs/cs.rs:243-245   brotli::Decompressor over Parameters.2
//   GateStreamedIterator::next     cs.rs:215-223   three parts per gate
//   read_gate_part                 cs.rs:193-213   u32 LE count | count x (32 B canonical LE Fr | u8 tag 0 = Input, 1 = Aux | u32 LE index)
// and, for writing a `Parameters` object (fk_gates_encode):
//   the writer                     backend/bellman_groth16/setup.rs:25-32   CompressorWriter(_, 4096, quality 9, lgwin 22) over
//                                  Gate::serialize (cs.rs:184-191: the three linear combinations, lc.rs:144-149)
// Coefficients arrive canonical and leave in Montgomery form (ff-uint_derive/src/lib.rs:696-701: values >= r are InvalidData);
// variables become Input(i) -> i, Aux(j) -> num_input + j (cs.rs:255-268).
//
// Decoding pipeline.  A brotli stream is one serial bit stream: nothing but the decompressor itself (libbrotlidec, ~1.5 GB/s of
// output on this data) can find byte k of the gate stream, so the decompressor is the floor and everything else is moved off its
// thread.  The decoding thread decompresses into blocks of a few MiB and WALKS ONLY THE COUNTS (a gate is self-delimiting: three
// u32 counts, each followed by count x 37 bytes) to cut the blocks at gate boundaries and to know, per block, the first gate and the
// offsets of its terms in the three matrices.  Worker threads parse the blocks side by side -- range checks, coefficient
// dictionary (a thread-private open-addressing table in front of a shared map: the benchmark's system has 8 126 distinct
// coefficients among 1.64e9 terms), structural density flags -- and write straight into the final arrays, which live in
// address space reserved up front and committed as the stream grows (no reallocation, no second copy: peak host memory is the
// 8 bytes per term the result needs).  The dictionary is renumbered to first-occurrence order at the end, so the result does
// not depend on the thread count.
//
// Brotli itself (RFC 7932; the reference uses the `brotli` crate) is the system's libbrotlidec.so.1 / libbrotlienc.so.1 -- present
// in this image and on the GPU box -- bound at run time with dlopen: the format's 122 KB static dictionary makes a private codec
// pointless.  Without the library FK_GATES_BROTLI fails loudly with FK_ERR_UNSUPPORTED; FK_GATES_RAW needs nothing.
#include "common.hpp"
#include <new>
#include <stdexcept>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <sched.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <unistd.h>
#include <unordered_map>

namespace fk {

std::string &tls_error() { static thread_local std::string e; return e; }      // what fk_last_error(NULL) returns

// Threads of a host-side fan-out.  Joined on EVERY path out of the scope -- an exception between two thread starts must not destroy a joinable
// std::thread (std::terminate inside an extern "C" entry point, which fk_guard cannot turn into a status code: ADVICE r5) -- and a share whose
// thread cannot be started (std::system_error: EAGAIN, the thread limit; or no memory for the vector) runs on the calling thread instead.
// Only for shares that terminate on their own (a fixed range, or a loop over a shared atomic counter).
struct ThreadSet {
    std::vector<std::thread> t;
    template <class F> void run(F f) {
        try { t.emplace_back(f); }
        catch (const std::exception &) { f(); }
    }
    void join() { for (auto &x : t) if (x.joinable()) x.join(); t.clear(); }
    ~ThreadSet() { join(); }
};

const RoctxApi &roctx_api() {
    static const RoctxApi api = [] {
        RoctxApi a;
        const char *e = getenv("FK_ROCTX");
        if (!e || !e[0] || e[0] == '0') return a;
        void *h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        a.pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!a.push || !a.pop) a = RoctxApi{};
        return a;
    }();
    return api;
}

// host threads this process may use: FK_HOST_THREADS, else the affinity mask capped by the cgroup CPU quota (a container can
// show 256 CPUs and be allowed the time of 16), at most 64
unsigned host_threads() {
    if (const char *e = getenv("FK_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)std::min(v, 256); }
    unsigned n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (!n) n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period) n = std::min<unsigned>(n, (unsigned)std::max(1ull, strtoull(q, nullptr, 10) / period));
        fclose(f);
    } else {
        long long quota = -1, period = 0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1ll, quota / period));
    }
    return std::min(std::max(n, 1u), 64u);
}

// A growable array that never moves: address space reserved once (PROT_NONE costs nothing), made readable / writable as the
// array grows.  Writers on other threads keep their pointers while the owner commits more.
struct Arena {
    uint8_t *base = nullptr;
    size_t reserved = 0, committed = 0;
    void *map = nullptr; size_t map_len = 0;
    static constexpr size_t STEP = (size_t)32 << 20, HUGE = (size_t)2 << 20;
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { release(); }
    bool reserve(size_t bytes) {
        release();
        bytes = (bytes + STEP - 1) / STEP * STEP + STEP;
        void *p = mmap(nullptr, bytes + HUGE, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return false;
        map = p; map_len = bytes + HUGE;
        base = (uint8_t *)(((uintptr_t)p + HUGE - 1) & ~(uintptr_t)(HUGE - 1)); reserved = bytes; committed = 0;
        // 2 MiB pages where the kernel offers them on request (transparent_hugepage = madvise): a first touch costs ~60 us on the
        // virtual machines this runs on, and with 4 KiB pages the parsing threads of a 13 GB result spend their time in the kernel
        // (measured: 17 s of system time per 0.9 GB of arrays; profiles/r05_gate_decode_hugepages.log)
        (void)madvise(base, reserved, MADV_HUGEPAGE);
        return true;
    }
    bool commit(size_t bytes) {
        if (bytes <= committed) return true;
        if (bytes > reserved) return false;
        const size_t want = std::min(reserved, (bytes + STEP - 1) / STEP * STEP);
        if (mprotect(base + committed, want - committed, PROT_READ | PROT_WRITE) != 0) return false;
        committed = want;
        return true;
    }
    void release() { if (map) munmap(map, map_len); map = nullptr; base = nullptr; map_len = reserved = committed = 0; }
    template <class T> T *as() const { return (T *)base; }
};

}  // namespace fk

struct fk_gates {
    uint32_t num_input = 0, num_aux = 0;
    uint64_t num_gates = 0;
    fk::Arena ptr[3], col[3], cidx[3];     // u64[num_gates + 1], u32[nnz], u32[nnz]
    uint64_t nnz[3] = {0, 0, 0};
    std::vector<fk::Fr> table;          // slot 0 = ONE
    std::vector<uint8_t> a_aux, b_in, b_aux;      // structural density flags (which variables the A / B side visits), bellman's DensityTracker
    uint64_t decoded_bytes = 0;
    // how the decoding went: wall seconds, seconds inside the decompressor, seconds the decoding thread waited for a free worker,
    // parse seconds summed over the workers, renumbering seconds, worker threads, blocks, compressed bytes
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

struct fk_blob { fk::Arena mem; size_t len = 0; double prof[4] = {0, 0, 0, 0}; };

namespace fk {

static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------ libbrotli bindings
struct BrotliDec {
    void *(*create)(void *, void *, void *) = nullptr;
    int (*stream)(void *, size_t *, const uint8_t **, size_t *, uint8_t **, size_t *) = nullptr;
    void (*destroy)(void *) = nullptr;
    bool ok = false;
    BrotliDec() {
        void *h = nullptr;
        for (const char *nm : {"libbrotlidec.so.1", "libbrotlidec.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        create = (decltype(create))dlsym(h, "BrotliDecoderCreateInstance");
        stream = (decltype(stream))dlsym(h, "BrotliDecoderDecompressStream");
        destroy = (decltype(destroy))dlsym(h, "BrotliDecoderDestroyInstance");
        ok = create && stream && destroy;
    }
};
static const BrotliDec &brotli_dec() { static BrotliDec a; return a; }

struct BrotliEnc {
    void *(*create)(void *, void *, void *) = nullptr;
    int (*set)(void *, int, uint32_t) = nullptr;
    int (*stream)(void *, int, size_t *, const uint8_t **, size_t *, uint8_t **, size_t *) = nullptr;
    int (*finished)(void *) = nullptr;
    void (*destroy)(void *) = nullptr;
    bool ok = false;
    BrotliEnc() {
        void *h = nullptr;
        for (const char *nm : {"libbrotlienc.so.1", "libbrotlienc.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        create = (decltype(create))dlsym(h, "BrotliEncoderCreateInstance");
        set = (decltype(set))dlsym(h, "BrotliEncoderSetParameter");
        stream = (decltype(stream))dlsym(h, "BrotliEncoderCompressStream");
        finished = (decltype(finished))dlsym(h, "BrotliEncoderIsFinished");
        destroy = (decltype(destroy))dlsym(h, "BrotliEncoderDestroyInstance");
        ok = create && set && stream && finished && destroy;
    }
};
static const BrotliEnc &brotli_enc() { static BrotliEnc a; return a; }

// ------------------------------------------------------------------------------------------ coefficient dictionary
struct Key32 {
    uint64_t w[4];
    bool operator==(const Key32 &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    uint64_t hash() const {
        uint64_t h = w[0] * 0x9e3779b97f4a7c15ull ^ w[1] * 0xc2b2ae3d27d4eb4full ^ w[2] * 0x165667b19e3779f9ull ^ w[3] * 0xd6e8feb86659fd93ull;
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        return h;
    }
};
struct Key32Hash { size_t operator()(const Key32 &k) const { return (size_t)k.hash(); } };

// the map all workers share (taken only on a worker's FIRST sight of a value): canonical bytes -> provisional index, the
// Montgomery table, and the position in the stream where the value was seen first (for the final renumbering)
struct SharedDict {
    std::mutex mu;
    std::unordered_map<Key32, uint32_t, Key32Hash> map;
    std::vector<Fr> table;
    std::vector<uint64_t> first_seen;
    // returns the provisional index, or -1 (non-canonical value) / -2 (table full)
    int64_t intern(const Key32 &k, uint64_t where) {
        std::lock_guard<std::mutex> lock(mu);
        auto it = map.find(k);
        if (it != map.end()) { if (where < first_seen[it->second]) first_seen[it->second] = where; return it->second; }
        Fr c; memcpy(&c, k.w, 32);
        for (int i = 7; i >= 0; i--) {           // canonical: strictly below r
            if (c.v[i] < FrParams::p(i)) break;
            if (c.v[i] > FrParams::p(i) || i == 0) return -1;
        }
        if (table.size() >= 0xffffffffull) return -2;
        const uint32_t idx = (uint32_t)table.size();
        map.emplace(k, idx);
        table.push_back(Fr::to_mont(c));
        first_seen.push_back(where);
        return idx;
    }
};

// a worker's private table: open addressing, grown by doubling
struct LocalDict {
    struct Slot { Key32 k; uint32_t idx; uint32_t used; };
    std::vector<Slot> slots;
    size_t count = 0;
    LocalDict() : slots(1024) { for (auto &s : slots) s.used = 0; }
    bool find(const Key32 &k, uint64_t h, uint32_t *idx) const {
        const size_t mask = slots.size() - 1;
        for (size_t i = h & mask;; i = (i + 1) & mask) {
            const Slot &s = slots[i];
            if (!s.used) return false;
            if (s.k == k) { *idx = s.idx; return true; }
        }
    }
    void put(const Key32 &k, uint64_t h, uint32_t idx) {
        if ((count + 1) * 2 > slots.size()) {
            std::vector<Slot> old; old.swap(slots);
            slots.resize(old.size() * 2); for (auto &s : slots) s.used = 0;
            count = 0;
            for (const Slot &s : old) if (s.used) put(s.k, s.k.hash(), s.idx);
        }
        const size_t mask = slots.size() - 1;
        size_t i = h & mask;
        while (slots[i].used) i = (i + 1) & mask;
        slots[i].k = k; slots[i].idx = idx; slots[i].used = 1; count++;
    }
};

// ------------------------------------------------------------------------------------------ the decoding pipeline
struct Block {
    const uint8_t *data = nullptr; size_t len = 0;      // whole gates only
    std::vector<uint8_t> own;                            // the bytes, when they came out of the decompressor
    uint64_t seq = 0, first_gate = 0, n_gates = 0, off[3] = {0, 0, 0};
};

struct Decoder {
    fk_gates *g;
    SharedDict dict;
    // queue: the decoding thread pushes, workers pop
    std::mutex mu;
    std::condition_variable cv_work, cv_room;
    std::deque<std::unique_ptr<Block>> queue;
    std::vector<std::vector<uint8_t>> pool;      // recycled block buffers (already faulted in)
    size_t in_flight = 0, max_in_flight = 4;
    bool closed = false;
    std::atomic<bool> failed{false};
    // the error of the EARLIEST block (what a serial parser would have hit first).  err_seq is read by the workers without the lock: a block is
    // skipped only when it lies BEHIND the earliest failure known so far -- a block in front of it is still parsed and may replace the recorded
    // error, so that the message and the FORMAT / OOM code do not depend on which worker got to which block first
    std::atomic<uint64_t> err_seq{~0ull}; int err_code = FK_ERR_FORMAT; std::string err;
    double parse_s = 0;

    explicit Decoder(fk_gates *g_) : g(g_) {
        const Fr one = Fr::one();
        Key32 k1{{1, 0, 0, 0}};
        dict.map.emplace(k1, 0u);
        dict.table.push_back(one);
        dict.first_seen.push_back(0);
    }

    void fail(uint64_t seq, int code, const char *msg) {
        std::lock_guard<std::mutex> lock(mu);
        if (seq < err_seq.load(std::memory_order_relaxed)) { err_seq.store(seq, std::memory_order_relaxed); err_code = code; try { err = msg; } catch (...) {} }
        failed.store(true);
        cv_room.notify_all();
    }

    // one worker: blocks in increasing order of seq (the queue is FIFO), so its first sight of a value is its earliest
    void worker() {
        LocalDict local;
        Key32 last{{0, 0, 0, 0}}; uint32_t last_idx = 0; bool have_last = false;
        { Key32 k1{{1, 0, 0, 0}}; local.put(k1, k1.hash(), 0); }
        uint64_t *const ptr[3] = {g->ptr[0].as<uint64_t>(), g->ptr[1].as<uint64_t>(), g->ptr[2].as<uint64_t>()};
        uint32_t *const col[3] = {g->col[0].as<uint32_t>(), g->col[1].as<uint32_t>(), g->col[2].as<uint32_t>()};
        uint32_t *const cix[3] = {g->cidx[0].as<uint32_t>(), g->cidx[1].as<uint32_t>(), g->cidx[2].as<uint32_t>()};
        uint8_t *const a_aux = g->a_aux.data(), *const b_in = g->b_in.data(), *const b_aux = g->b_aux.data();
        const uint32_t n_in = g->num_input, n_aux = g->num_aux;
        double busy = 0;
        for (;;) {
            std::unique_ptr<Block> b;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_work.wait(lock, [&] { return !queue.empty() || closed; });
                if (queue.empty()) break;
                b = std::move(queue.front()); queue.pop_front();
            }
            const double t0 = now_s();
            if (b->seq < err_seq.load(std::memory_order_relaxed)) try {
                const uint8_t *p = b->data;
                uint64_t run[3] = {b->off[0], b->off[1], b->off[2]};
                uint64_t ordinal = 0;
                const char *bad = nullptr;
                for (uint64_t gi = 0; gi < b->n_gates && !bad; gi++) {
                    for (int k = 0; k < 3 && !bad; k++) {
                        uint32_t cnt; memcpy(&cnt, p, 4); p += 4;
                        uint64_t o = run[k];
                        for (uint32_t i = 0; i < cnt; i++, p += 37, o++, ordinal++) {
                            const uint8_t tag = p[32];
                            uint32_t idx; memcpy(&idx, p + 33, 4);
                            uint32_t v;
                            if (tag == 0) { if (idx >= n_in) { bad = "input index out of range"; break; } v = idx; }
                            else if (tag == 1) { if (idx >= n_aux) { bad = "aux index out of range"; break; } v = n_in + idx; }
                            else { bad = "enum elements overflow"; break; }                  // cs.rs:209
                            Key32 key; memcpy(key.w, p, 32);
                            uint32_t ci;
                            if (have_last && key == last) ci = last_idx;
                            else {
                                const uint64_t h = key.hash();
                                if (!local.find(key, h, &ci)) {
                                    const int64_t r = dict.intern(key, b->seq << 36 | ordinal);
                                    if (r < 0) { bad = r == -1 ? "non-canonical field element" : "too many distinct coefficients"; break; }
                                    ci = (uint32_t)r;
                                    local.put(key, h, ci);
                                }
                                last = key; last_idx = ci; have_last = true;
                            }
                            col[k][o] = v; cix[k][o] = ci;
                            // bellman's DensityTracker: a variable is counted when an A- / B-side combination visits it, whatever its value
                            // (every writer stores the same 1: relaxed atomic stores, plain byte moves in the generated code)
                            if (k == 0) { if (tag) __atomic_store_n(&a_aux[idx], 1, __ATOMIC_RELAXED); }
                            else if (k == 1) { if (tag) __atomic_store_n(&b_aux[idx], 1, __ATOMIC_RELAXED); else __atomic_store_n(&b_in[idx], 1, __ATOMIC_RELAXED); }
                        }
                        run[k] = o;
                        ptr[k][b->first_gate + gi + 1] = o;
                    }
                }
                if (bad) fail(b->seq, FK_ERR_FORMAT, bad);
            } catch (const std::exception &) { fail(b->seq, FK_ERR_OOM, "out of host memory while decoding the gate stream"); }      // (an exception must not leave a thread)
            busy += now_s() - t0;
            {
                std::lock_guard<std::mutex> lock(mu);
                if (!b->own.empty() && pool.size() < 64) pool.push_back(std::move(b->own));
                in_flight--;
            }
            cv_room.notify_one();
        }
        std::lock_guard<std::mutex> lock(mu);
        parse_s += busy;
    }

    // the decoding thread hands a block of whole gates to the workers (waits while too many are in flight); false once a worker failed
    bool push(std::unique_ptr<Block> b, double *waited) {
        std::unique_lock<std::mutex> lock(mu);
        if (in_flight >= max_in_flight && !failed.load()) {
            const double t0 = now_s();
            cv_room.wait(lock, [&] { return in_flight < max_in_flight || failed.load(); });
            *waited += now_s() - t0;
        }
        if (failed.load()) return false;
        in_flight++;
        queue.push_back(std::move(b));
        lock.unlock();
        cv_work.notify_one();
        return true;
    }
    std::vector<uint8_t> buffer(size_t cap) {
        std::vector<uint8_t> v;
        { std::lock_guard<std::mutex> lock(mu); if (!pool.empty()) { v = std::move(pool.back()); pool.pop_back(); } }
        if (v.size() < cap) v.resize(cap);
        return v;
    }
    void close() { { std::lock_guard<std::mutex> lock(mu); closed = true; } cv_work.notify_all(); }
};

// walks the counts of a gate stream: where do whole gates end, how many terms does each matrix get
struct Scanner {
    uint64_t num_gates, gates = 0, nnz[3] = {0, 0, 0};
    size_t pos = 0, gate_end = 0;          // next count field not yet consumed; end of the last whole gate (both relative to the current buffer)
    int part = 0;
    uint64_t pend[3] = {0, 0, 0};          // terms of the parts of the gate in progress
    // the block under construction
    uint64_t blk_gates = 0, blk_nnz[3] = {0, 0, 0};
    explicit Scanner(uint64_t n) : num_gates(n) {}
    // advances over buf[0, fill); stops at the first incomplete part, or once `stop_at` bytes of whole gates are behind it.
    // false: bytes behind the last gate
    bool scan(const uint8_t *buf, size_t fill, size_t stop_at = ~(size_t)0) {
        for (;;) {
            if (gates == num_gates) return pos == fill;
            if (gate_end >= stop_at && part == 0) return true;
            if (pos + 4 > fill) return true;
            uint32_t cnt; memcpy(&cnt, buf + pos, 4);
            const uint64_t need = 4 + (uint64_t)cnt * 37;
            if (need > fill - pos) return true;
            pos += need; pend[part] = cnt;
            if (++part == 3) {
                part = 0; gates++; gate_end = pos; blk_gates++;
                for (int k = 0; k < 3; k++) blk_nnz[k] += pend[k];
            }
        }
    }
    // bytes a part in progress still needs beyond `fill` (so that a giant linear combination grows the buffer once)
    size_t shortfall(const uint8_t *buf, size_t fill) const {
        if (gates == num_gates || pos + 4 > fill) return 0;
        uint32_t cnt; memcpy(&cnt, buf + pos, 4);
        const uint64_t need = 4 + (uint64_t)cnt * 37;
        return need > fill - pos ? (size_t)(need - (fill - pos)) : 0;
    }
};

static int gates_decode(fk_ctx *ctx, const uint8_t *blob, size_t len, int format, uint32_t num_gates, uint32_t num_input, uint32_t num_aux, fk_gates **out) {
    if (!out || (len && !blob)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: null argument");
    *out = nullptr;
    if (num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: num_input must include the constant ONE");
    if ((uint64_t)num_input + num_aux > 0xffffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: too many variables");
    if (format != FK_GATES_RAW && format != FK_GATES_BROTLI) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: unknown blob format %d", format);
    // every gate is at least three 4-byte counts: a header that promises more gates than the stream can hold is malformed
    if (format == FK_GATES_RAW && (uint64_t)num_gates * 12 > len) FK_SET_ERR(ctx, FK_ERR_FORMAT, "gates: %u gates cannot fit into %zu bytes", num_gates, len);
    const double t_start = now_s();
    std::unique_ptr<fk_gates> gp(new fk_gates());
    fk_gates *g = gp.get();
    g->num_input = num_input; g->num_aux = num_aux; g->num_gates = num_gates;
    auto fail = [&](int code, const std::string &msg) { ctx->err = "gates: " + msg; return code; };
    // num_gates comes straight from a file header: address space for what it promises (PROT_NONE: free), memory only for what arrives
    // (2^36 terms per matrix; a process with a small address-space limit -- ulimit -v -- gets what it can reserve, down to 2^22 terms, and a
    // stream that outgrows it is FK_ERR_OOM when it does; a raw stream says how many terms it can hold at most)
    size_t term_space = (size_t)1 << 38;
    if (format == FK_GATES_RAW) term_space = std::min(term_space, std::max<size_t>(len / 37 * 4 + 64, (size_t)1 << 20));
    for (;;) {
        bool ok = true;
        for (int k = 0; k < 3 && ok; k++) ok = g->ptr[k].reserve(((size_t)num_gates + 1) * 8) && g->col[k].reserve(term_space) && g->cidx[k].reserve(term_space);
        if (ok) break;
        if (term_space <= ((size_t)1 << 24)) return fail(FK_ERR_OOM, "cannot reserve address space for the constraint system");
        term_space >>= 2;
    }
    for (int k = 0; k < 3; k++) {
        if (!g->ptr[k].commit(8)) return fail(FK_ERR_OOM, "out of host memory");
        g->ptr[k].as<uint64_t>()[0] = 0;
    }
    g->a_aux.assign(num_aux ? num_aux : 1, 0); g->b_aux.assign(num_aux ? num_aux : 1, 0); g->b_in.assign(num_input, 0);

    const BrotliDec &br = brotli_dec();
    void *st = nullptr;
    if (format == FK_GATES_BROTLI) {
        if (!br.ok) return fail(FK_ERR_UNSUPPORTED, "libbrotlidec.so.1 not found (needed for a brotli gate blob)");
        st = br.create(nullptr, nullptr, nullptr);
        if (!st) return fail(FK_ERR_OOM, "brotli decoder allocation failed");
    }
    struct StGuard { void *st; const BrotliDec *br; ~StGuard() { if (st) br->destroy(st); } } st_guard{st, &br};

    Decoder dec(g);
    const unsigned n_host = host_threads();
    const unsigned n_workers = std::max(1u, n_host - (format == FK_GATES_BROTLI && n_host > 2 ? 1u : 0u));      // the decompressor keeps a thread to itself
    dec.max_in_flight = 2 * n_workers + 2;
    std::vector<std::thread> threads;
    struct Joiner { Decoder &d; std::vector<std::thread> &t; ~Joiner() { d.close(); for (auto &x : t) if (x.joinable()) x.join(); } } joiner{dec, threads};
    // small inputs are parsed by one worker: no point in starting 16 threads for a 7 000-gate circuit
    const bool small = format == FK_GATES_RAW ? len < ((size_t)4 << 20) : len < ((size_t)256 << 10);
    const unsigned n_start = small ? 1 : n_workers;
    for (unsigned i = 0; i < n_start; i++) threads.emplace_back([&dec] { dec.worker(); });

    Scanner sc(num_gates);
    uint64_t seq = 0, first_gate = 0, off[3] = {0, 0, 0};
    double t_brotli = 0, t_wait = 0, t_scan = 0, t_buf = 0, t_disp = 0;
    int rc = FK_OK; std::string msg;
    // hands buf[0, sc.gate_end) to the workers as one block (commits the arrays it will write first)
    auto dispatch = [&](const uint8_t *data, std::vector<uint8_t> *own) -> bool {
        if (!sc.blk_gates) return true;
        for (int k = 0; k < 3; k++) {
            if (!g->ptr[k].commit((first_gate + sc.blk_gates + 1) * 8) || !g->col[k].commit((off[k] + sc.blk_nnz[k]) * 4 + 4) || !g->cidx[k].commit((off[k] + sc.blk_nnz[k]) * 4 + 4)) {
                rc = FK_ERR_OOM; msg = "out of host memory (or of the address space this process may reserve) while decoding the gate stream"; return false;
            }
        }
        if (seq >= ((uint64_t)1 << 27)) { rc = FK_ERR_FORMAT; msg = "gate stream too long"; return false; }
        std::unique_ptr<Block> b(new Block());
        b->seq = seq++; b->first_gate = first_gate; b->n_gates = sc.blk_gates; b->len = sc.gate_end;
        for (int k = 0; k < 3; k++) { b->off[k] = off[k]; off[k] += sc.blk_nnz[k]; sc.blk_nnz[k] = 0; }
        first_gate += sc.blk_gates; sc.blk_gates = 0;
        if (own) { b->own = std::move(*own); b->data = b->own.data(); } else b->data = data;
        return dec.push(std::move(b), &t_wait);
    };

    if (format == FK_GATES_RAW) {
        // the stream is in memory: cut it into blocks in place
        const size_t TARGET = (size_t)8 << 20;
        g->decoded_bytes = len;
        size_t base = 0;
        for (;;) {
            if (!sc.scan(blob + base, len - base, TARGET)) { rc = FK_ERR_FORMAT; msg = "trailing bytes after the last gate"; break; }
            if (!sc.blk_gates) break;               // no whole gate left
            const size_t end = sc.gate_end;
            if (!dispatch(blob + base, nullptr)) break;
            base += end; sc.pos -= end; sc.gate_end = 0;
        }
        if (rc == FK_OK && !dec.failed.load() && (base != len || sc.gates != num_gates)) { rc = FK_ERR_FORMAT; msg = "gate stream truncated (fewer than num_gates gates)"; }
    } else {
        // The decompressor is asked for 256 KiB of output at a time: the scanner that follows every call is a POINTER CHASE through the bytes just
        // written (a count says where the next count is), 10^8 dependent loads for the benchmark's stream -- with 1 MiB pieces they had left the
        // near caches by the time the scanner came (3.9 s of a 16.2 s decode), with 256 KiB they have not (0.3 s; 64 KiB the same, 16 KiB costs the
        // decompressor more calls than it saves: profiles/r05_decode_chunk_probe.log).  FK_GATES_CHUNK_KB overrides.
        static const size_t chunk_kb = getenv("FK_GATES_CHUNK_KB") ? (size_t)std::min(8192, std::max(4, atoi(getenv("FK_GATES_CHUNK_KB")))) : 256;
        const size_t TARGET = (size_t)8 << 20, CHUNK = chunk_kb << 10;
        std::vector<uint8_t> cur = dec.buffer(TARGET + 2 * CHUNK);
        size_t fill = 0;
        size_t avail_in = len; const uint8_t *next_in = blob;
        int res = 3;
        while (rc == FK_OK) {
            // room for the next piece of output (a linear combination larger than a block grows the buffer once, to its size)
            const size_t want = fill + CHUNK + sc.shortfall(cur.data(), fill);
            if (cur.size() < want) cur.resize(std::max(want, cur.size() + cur.size() / 2));
            size_t avail_out = std::min(cur.size() - fill, std::max(CHUNK, sc.shortfall(cur.data(), fill)));
            uint8_t *next_out = cur.data() + fill;
            const size_t before = avail_out;
            const double t0 = now_s();
            res = br.stream(st, &avail_in, &next_in, &avail_out, &next_out, nullptr);     // 0 error, 1 done, 2 needs input, 3 needs output
            t_brotli += now_s() - t0;
            if (res == 0) { rc = FK_ERR_FORMAT; msg = "corrupt brotli stream"; break; }
            fill += before - avail_out;
            g->decoded_bytes += before - avail_out;
            const double ts0 = now_s();
            const bool sok = sc.scan(cur.data(), fill);
            t_scan += now_s() - ts0;
            if (!sok) { rc = FK_ERR_FORMAT; msg = "trailing bytes after the last gate"; break; }
            if (res == 2 && avail_in == 0) { rc = FK_ERR_FORMAT; msg = "brotli stream truncated"; break; }
            if (sc.gate_end >= TARGET || (res == 1 && sc.blk_gates)) {
                const double td0 = now_s();
                const size_t end = sc.gate_end, tail = fill - end;
                std::vector<uint8_t> nxt = dec.buffer(std::max(TARGET + 2 * CHUNK, tail + CHUNK));
                if (tail) memcpy(nxt.data(), cur.data() + end, tail);
                const double td1 = now_s();
                const bool dok = dispatch(nullptr, &cur);
                t_buf += td1 - td0; t_disp += now_s() - td1;
                if (!dok) break;
                cur = std::move(nxt); fill = tail; sc.pos -= end; sc.gate_end = 0;
            }
            if (res == 1) {
                if (fill || sc.gates != num_gates) { rc = FK_ERR_FORMAT; msg = "gate stream truncated (fewer than num_gates gates)"; }
                break;
            }
        }
    }
    const double t_join = now_s();
    dec.close();
    for (auto &x : threads) if (x.joinable()) x.join();
    if (dec.failed.load()) return fail(dec.err_code, dec.err);          // the earliest block's error: what a serial reader would have met
    if (rc != FK_OK) return fail(rc, msg);
    if (sc.gates != num_gates) return fail(FK_ERR_FORMAT, "gate stream truncated (fewer than num_gates gates)");
    for (int k = 0; k < 3; k++) g->nnz[k] = off[k];

    // renumber the dictionary to first-occurrence order (what a serial reader assigns): independent of thread count and timing
    const double t_ren = now_s();
    const size_t nt = dec.dict.table.size();
    std::vector<uint32_t> order(nt), perm(nt);
    for (size_t i = 0; i < nt; i++) order[i] = (uint32_t)i;
    std::sort(order.begin() + 1, order.end(), [&](uint32_t a, uint32_t b) { return dec.dict.first_seen[a] < dec.dict.first_seen[b]; });
    bool identity = true;
    g->table.resize(nt);
    for (size_t i = 0; i < nt; i++) { perm[order[i]] = (uint32_t)i; g->table[i] = dec.dict.table[order[i]]; if (order[i] != i) identity = false; }
    if (!identity) {
        const unsigned nth = small ? 1 : n_workers;
        ThreadSet rt;
        for (int k = 0; k < 3; k++) {
            uint32_t *c = g->cidx[k].as<uint32_t>();
            const uint64_t n = g->nnz[k];
            for (unsigned t = 0; t < nth; t++)
                rt.run([=, &perm] { for (uint64_t i = n * t / nth, e = n * (t + 1) / nth; i < e; i++) c[i] = perm[c[i]]; });
        }
        rt.join();
    }
    const double t_end = now_s();
    if (getenv("FK_GATES_TRACE")) fprintf(stderr, "[fk] gates: wall %.2f brotli %.2f scan %.2f buffers %.2f dispatch %.2f (of it waiting %.2f) join+renumber %.2f\n", t_end - t_start, t_brotli, t_scan, t_buf, t_disp, t_wait, t_end - t_join);
    const double prof[8] = {t_end - t_start, t_brotli, t_wait, dec.parse_s, t_end - t_ren, (double)n_start, (double)seq, (double)len};
    memcpy(g->prof, prof, sizeof prof);
    *out = gp.release();
    return FK_OK;
}

// ------------------------------------------------------------------------------------------ the encoder
// Gate::serialize of every gate of `copies` copies of `cs` (fk_r1cs_load_tiled's variable order: ONE shared, copy j's inputs at
// 1 + j * (num_input - 1), its aux variables at j * num_aux), formatted in units of a few MiB by worker threads and compressed, in
// order, by the calling thread.  With copies > 1 ONE copy's stream is kept as a template (canonical coefficients, tags) and a unit
// is a memcpy plus its index fields; a single explicit system is formatted straight from the CSR.
static int gates_encode(fk_ctx *ctx, const fk_r1cs *cs, uint32_t copies, int format, int quality, int lgwin, fk_blob **out) {
    if (!cs || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: null argument");
    *out = nullptr;
    if (copies == 0 || cs->num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: copies and num_input must be at least 1");
    if (format != FK_GATES_RAW && format != FK_GATES_BROTLI) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: unknown blob format %d", format);
    if (format == FK_GATES_BROTLI && (quality < 0 || quality > 11 || lgwin < 10 || lgwin > 24)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: brotli quality 0 .. 11, lgwin 10 .. 24");
    const uint64_t *ptrs[3] = {cs->a_ptr, cs->b_ptr, cs->c_ptr};
    const uint32_t *cols[3] = {cs->a_col, cs->b_col, cs->c_col};
    const uint64_t *vals[3] = {cs->a_val, cs->b_val, cs->c_val};
    const uint64_t G = cs->num_gates, nv = (uint64_t)cs->num_input + cs->num_aux;
    const uint64_t t_in = 1 + (uint64_t)copies * (cs->num_input - 1), t_aux = (uint64_t)copies * cs->num_aux;
    if (t_in + t_aux > 0xffffffffull || G * copies > 0xffffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: %u copies of this system do not fit 32-bit indices", copies);
    for (int k = 0; k < 3; k++) {
        if (!ptrs[k] || ptrs[k][0] != 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: bad row pointer");
        for (uint64_t gi = 0; gi < G; gi++) {
            if (ptrs[k][gi + 1] < ptrs[k][gi]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: row_ptr not monotone");
            if (ptrs[k][gi + 1] - ptrs[k][gi] > 0xffffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: a linear combination with more than 2^32 terms");
        }
        if (ptrs[k][G] && !cols[k]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: null column array");
        for (uint64_t i = 0; i < ptrs[k][G]; i++) if (cols[k][i] >= nv) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: variable index %u out of range", cols[k][i]);
    }
    const double t_start = now_s();
    std::vector<uint64_t> gate_off(G + 1, 0);          // byte offset of gate gi in one copy's stream
    for (uint64_t gi = 0; gi < G; gi++) {
        uint64_t sz = 12;
        for (int k = 0; k < 3; k++) sz += 37 * (ptrs[k][gi + 1] - ptrs[k][gi]);
        gate_off[gi + 1] = gate_off[gi] + sz;
    }
    const uint64_t copy_bytes = gate_off[G];
    // units: runs of whole gates of at most ~8 MiB (a single gate may be larger)
    std::vector<uint64_t> unit_g{0};
    for (uint64_t gi = 0; gi < G;) {
        uint64_t e = gi + 1;
        while (e < G && gate_off[e + 1] - gate_off[gi] <= ((uint64_t)8 << 20)) e++;
        unit_g.push_back(e); gi = e;
    }
    const uint64_t R = unit_g.size() - 1, n_units = R * copies;
    uint64_t max_unit = 1;
    for (uint64_t u = 0; u < R; u++) max_unit = std::max(max_unit, gate_off[unit_g[u + 1]] - gate_off[unit_g[u]]);

    // formats the gates [g0, g1) of copy j at dst, straight from the CSR
    auto format_range = [&](uint32_t j, uint64_t g0, uint64_t g1, uint8_t *dst) {
        const uint32_t d_in = j * (cs->num_input - 1), d_aux = j * cs->num_aux;
        static const uint8_t one_c[32] = {1};
        const Fr one = Fr::one();
        Fr last = one; uint8_t last_c[32]; memcpy(last_c, one_c, 32);
        uint8_t *p = dst;
        for (uint64_t gi = g0; gi < g1; gi++)
            for (int k = 0; k < 3; k++) {
                const uint64_t lo = ptrs[k][gi], hi = ptrs[k][gi + 1];
                const uint32_t cnt = (uint32_t)(hi - lo);
                memcpy(p, &cnt, 4); p += 4;
                for (uint64_t i = lo; i < hi; i++, p += 37) {
                    if (!vals[k]) memcpy(p, one_c, 32);
                    else {
                        if (memcmp(vals[k] + 4 * i, &last, 32) != 0) { memcpy(&last, vals[k] + 4 * i, 32); const Fr c = Fr::from_mont(last); memcpy(last_c, &c, 32); }
                        memcpy(p, last_c, 32);
                    }
                    const uint32_t v = cols[k][i];
                    uint32_t idx; uint8_t tag;
                    if (v < cs->num_input) { tag = 0; idx = v ? v + d_in : 0; }
                    else { tag = 1; idx = v - cs->num_input + d_aux; }
                    p[32] = tag; memcpy(p + 33, &idx, 4);
                }
            }
    };
    // copies > 1: copy 0's stream once, and where the index fields of a copy's own variables sit in it
    struct Patch { uint64_t at; uint32_t base; uint32_t kind; };      // kind 1: a copy's input, 2: a copy's aux (ONE keeps index 0)
    std::vector<uint8_t> tpl;
    std::vector<Patch> patches;
    std::vector<uint64_t> unit_patch;
    if (copies > 1) {
        tpl.resize(copy_bytes ? copy_bytes : 1);
        {
            const unsigned nth = (unsigned)std::min<uint64_t>(host_threads(), R ? R : 1);
            std::atomic<uint64_t> next{0};
            ThreadSet th;
            for (unsigned t = 0; t < nth; t++) th.run([&] { for (uint64_t u; (u = next.fetch_add(1)) < R;) format_range(0, unit_g[u], unit_g[u + 1], tpl.data() + gate_off[unit_g[u]]); });
            th.join();
        }
        unit_patch.assign(R + 1, 0);
        uint64_t at = 0;
        for (uint64_t u = 0; u < R; u++) {
            for (uint64_t gi = unit_g[u]; gi < unit_g[u + 1]; gi++)
                for (int k = 0; k < 3; k++) {
                    at += 4;
                    for (uint64_t i = ptrs[k][gi]; i < ptrs[k][gi + 1]; i++, at += 37) {
                        const uint32_t v = cols[k][i];
                        if (v == 0) continue;
                        if (v < cs->num_input) patches.push_back({at + 33, v, 1u}); else patches.push_back({at + 33, v - cs->num_input, 2u});
                    }
                }
            unit_patch[u + 1] = patches.size();
        }
    }
    auto fill_unit = [&](uint64_t unit, uint8_t *dst) {
        const uint32_t j = (uint32_t)(unit / R); const uint64_t u = unit % R;
        if (copies == 1) { format_range(0, unit_g[u], unit_g[u + 1], dst); return; }
        const uint64_t o = gate_off[unit_g[u]];
        memcpy(dst, tpl.data() + o, gate_off[unit_g[u + 1]] - o);
        if (j == 0) return;
        const uint32_t d_in = j * (cs->num_input - 1), d_aux = j * cs->num_aux;
        for (uint64_t q = unit_patch[u]; q < unit_patch[u + 1]; q++) { const Patch &pt = patches[q]; const uint32_t idx = pt.base + (pt.kind == 1 ? d_in : d_aux); memcpy(dst + (pt.at - o), &idx, 4); }
    };
    auto unit_bytes = [&](uint64_t unit) { const uint64_t u = unit % R; return gate_off[unit_g[u + 1]] - gate_off[unit_g[u]]; };

    std::unique_ptr<fk_blob> bp(new fk_blob());
    fk_blob *bl = bp.get();
    const unsigned __int128 total = (unsigned __int128)copy_bytes * copies;
    if (total >> 46) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: a gate stream of more than 2^46 bytes");
    if (!bl->mem.reserve(format == FK_GATES_RAW ? (size_t)total + 64 : (size_t)(total + total / 256) + ((size_t)64 << 20))) FK_SET_ERR(ctx, FK_ERR_OOM, "gates: cannot reserve address space for the blob");
    if (format == FK_GATES_RAW) {
        if (!bl->mem.commit((size_t)total + 1)) FK_SET_ERR(ctx, FK_ERR_OOM, "gates: out of host memory for the raw gate stream");
        const unsigned nth = (unsigned)std::min<uint64_t>(host_threads(), n_units ? n_units : 1);
        std::atomic<uint64_t> next{0};
        ThreadSet th;
        for (unsigned t = 0; t < nth; t++) th.run([&] {
            for (uint64_t unit; (unit = next.fetch_add(1)) < n_units;) fill_unit(unit, bl->mem.base + (size_t)(unit / R) * copy_bytes + gate_off[unit_g[unit % R]]);
        });
        th.join();
        bl->len = (size_t)total;
    } else {
        const BrotliEnc &be = brotli_enc();
        if (!be.ok) FK_SET_ERR(ctx, FK_ERR_UNSUPPORTED, "gates: libbrotlienc.so.1 not found (needed to write a brotli gate blob)");
        void *st = be.create(nullptr, nullptr, nullptr);
        if (!st) FK_SET_ERR(ctx, FK_ERR_OOM, "gates: brotli encoder allocation failed");
        struct G_ { void *st; const BrotliEnc *be; ~G_() { be->destroy(st); } } guard{st, &be};
        be.set(st, 1, (uint32_t)quality);          // BROTLI_PARAM_QUALITY
        be.set(st, 2, (uint32_t)lgwin);            // BROTLI_PARAM_LGWIN
        // formatting threads fill a ring of unit buffers ahead of the compressor
        const unsigned n_host = host_threads();
        const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_host > 1 ? n_host - 1 : 1, n_units));
        const uint64_t ring = std::max<uint64_t>(1, std::min<uint64_t>(n_units, 2 * nth + 2));
        std::vector<std::vector<uint8_t>> bufs(ring);
        for (auto &b : bufs) b.resize(max_unit);
        std::mutex mu; std::condition_variable cv;
        std::vector<int64_t> holds(ring, -1);          // which unit a ring slot holds, once formatted
        uint64_t consumed = 0, claimed = 0;            // units compressed so far; units handed to a formatting thread
        bool stop = false;
        std::vector<std::thread> th;
        th.reserve(nth);
        // (declared BEFORE the first thread starts: whatever leaves this scope -- a status code, an exception between two thread starts -- first
        // tells the formatters to stop and joins them; a joinable std::thread must never be destroyed: std::terminate inside an extern "C" entry)
        struct J_ { std::mutex &mu; std::condition_variable &cv; bool &stop; std::vector<std::thread> &th; ~J_() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); for (auto &x : th) if (x.joinable()) x.join(); } } joiner{mu, cv, stop, th};
        for (unsigned t = 0; t < nth; t++) try { th.emplace_back([&] {
            for (;;) {
                uint64_t unit;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return stop || claimed >= n_units || claimed < consumed + ring; });
                    if (stop || claimed >= n_units) return;
                    unit = claimed++;
                }
                fill_unit(unit, bufs[unit % ring].data());
                { std::lock_guard<std::mutex> lock(mu); holds[unit % ring] = (int64_t)unit; }
                cv.notify_all();
            }
        }); } catch (const std::system_error &) { break; }        // the thread limit: go on with the formatters that did start
        if (th.empty()) FK_SET_ERR(ctx, FK_ERR_OOM, "gates: no formatting thread could be started");
        size_t out_len = 0;
        double t_enc = 0;
        auto pump = [&](int op, const uint8_t *src, size_t n) -> int {
            size_t avail_in = n; const uint8_t *next_in = src;
            for (;;) {
                if (!bl->mem.commit(out_len + ((size_t)4 << 20))) FK_SET_ERR(ctx, FK_ERR_OOM, "gates: out of host memory for the blob");
                size_t avail_out = bl->mem.committed - out_len; uint8_t *next_out = bl->mem.base + out_len;
                const size_t before = avail_out;
                const double t0 = now_s();
                const int ok = be.stream(st, op, &avail_in, &next_in, &avail_out, &next_out, nullptr);
                t_enc += now_s() - t0;
                if (!ok) FK_SET_ERR(ctx, FK_ERR_HIP, "gates: brotli encoder failed");
                out_len += before - avail_out;
                if (avail_in == 0 && (op != 2 ? avail_out != 0 : be.finished(st) != 0)) return FK_OK;      // BROTLI_OPERATION_FINISH = 2
            }
        };
        for (uint64_t unit = 0; unit < n_units; unit++) {
            { std::unique_lock<std::mutex> lock(mu); cv.wait(lock, [&] { return holds[unit % ring] == (int64_t)unit; }); }
            FK_TRY(pump(0, bufs[unit % ring].data(), unit_bytes(unit)));          // BROTLI_OPERATION_PROCESS
            { std::lock_guard<std::mutex> lock(mu); consumed = unit + 1; }
            cv.notify_all();
        }
        FK_TRY(pump(2, nullptr, 0));
        bl->len = out_len;
        bl->prof[1] = t_enc;
    }
    bl->prof[0] = now_s() - t_start; bl->prof[2] = (double)total; bl->prof[3] = (double)bl->len;
    *out = bp.release();
    return FK_OK;
}

}  // namespace fk

using namespace fk;

extern "C" {

// 1 when roctx ranges are being emitted (FK_ROCTX=1 and libroctx64 bound), else 0
int fk_roctx_active(void) { return fk::roctx_api().push != nullptr ? 1 : 0; }

int fk_gates_decode(fk_ctx *ctx, const uint8_t *blob, size_t len, int format, uint32_t num_gates, uint32_t num_input, uint32_t num_aux, fk_gates **out) {
    FK_RANGE("fk_gates_decode");
    fk_ctx local;                  // host-only routine: usable without a GPU context
    if (!ctx) ctx = &local;
    // a tiny brotli blob can inflate to anything: running out of host memory is a status code, never an exception that leaves
    // an extern "C" function (std::terminate would take the ctypes / Rust host down)
    const int rc = fk_guard(ctx, [&]() -> int { return gates_decode(ctx, blob, len, format, num_gates, num_input, num_aux, out); });
    if (rc != FK_OK && out) *out = nullptr;
    if (ctx == &local) { try { tls_error() = local.err; } catch (...) {} }
    return rc;
}

void fk_gates_free(fk_gates *g) { delete g; }

int fk_gates_info(const fk_gates *g, uint64_t out[8]) {
    if (!g || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[8] = {g->num_gates, g->nnz[0], g->nnz[1], g->nnz[2], g->table.size(), g->decoded_bytes, g->num_input, g->num_aux};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

int fk_gates_profile(const fk_gates *g, double out[8]) {
    if (!g || !out) return FK_ERR_BAD_ARG;
    memcpy(out, g->prof, sizeof g->prof);
    return FK_OK;
}

// one matrix as the arrays of an fk_r1cs: ptr[num_gates + 1], col[nnz], val[nnz x 4] (Montgomery; may be NULL)
int fk_gates_export(const fk_gates *g, int mtx, uint64_t *ptr, uint32_t *col, uint64_t *val) {
    if (!g || mtx < 0 || mtx > 2 || !ptr) return FK_ERR_BAD_ARG;
    memcpy(ptr, g->ptr[mtx].base, (g->num_gates + 1) * 8);
    const size_t nnz = g->nnz[mtx];
    if (col && nnz) memcpy(col, g->col[mtx].base, nnz * 4);
    const uint32_t *ci = g->cidx[mtx].as<uint32_t>();
    if (val) for (size_t i = 0; i < nnz; i++) memcpy(val + 4 * i, &g->table[ci[i]], 32);
    return FK_OK;
}

int fk_gates_encode(fk_ctx *ctx, const fk_r1cs *cs, uint32_t copies, int format, int quality, int lgwin, fk_blob **out) {
    fk_ctx local;
    if (!ctx) ctx = &local;
    const int rc = fk_guard(ctx, [&]() -> int { return gates_encode(ctx, cs, copies, format, quality, lgwin, out); });
    if (rc != FK_OK && out) *out = nullptr;
    if (ctx == &local) { try { tls_error() = local.err; } catch (...) {} }
    return rc;
}

int fk_blob_data(const fk_blob *b, const uint8_t **data, size_t *len) {
    if (!b || !data || !len) return FK_ERR_BAD_ARG;
    *data = b->mem.base; *len = b->len;
    return FK_OK;
}

int fk_blob_profile(const fk_blob *b, double out[4]) {
    if (!b || !out) return FK_ERR_BAD_ARG;
    memcpy(out, b->prof, sizeof b->prof);
    return FK_OK;
}

void fk_blob_free(fk_blob *b) { delete b; }

}  // extern "C"

// the resident constraint system straight from the decoded stream (spmv.hip); its dictionary and density flags are reused as they are
namespace fk { int r1cs_load_coded(fk_ctx *ctx, uint32_t num_input, uint32_t num_aux, uint64_t num_gates, const uint64_t *const ptr[3], const uint32_t *const col[3],
                                   const uint32_t *const cidx[3], const Fr *table, uint64_t n_table, fk_r1cs_dev **out, const uint8_t *const density[3] = nullptr); }

extern "C" int fk_r1cs_load_gates(fk_ctx *ctx, const fk_gates *g, fk_r1cs_dev **out) {
    if (!ctx || !g || !out) return FK_ERR_BAD_ARG;
    return fk_guard(ctx, [&]() -> int {
        const uint64_t *ptr[3] = {g->ptr[0].as<uint64_t>(), g->ptr[1].as<uint64_t>(), g->ptr[2].as<uint64_t>()};
        const uint32_t *col[3] = {g->col[0].as<uint32_t>(), g->col[1].as<uint32_t>(), g->col[2].as<uint32_t>()};
        const uint32_t *cidx[3] = {g->cidx[0].as<uint32_t>(), g->cidx[1].as<uint32_t>(), g->cidx[2].as<uint32_t>()};
        const uint8_t *dens[3] = {g->a_aux.data(), g->b_in.data(), g->b_aux.data()};
        return r1cs_load_coded(ctx, g->num_input, g->num_aux, g->num_gates, ptr, col, cidx, g->table.data(), g->table.size(), out, dens);
    });
}
