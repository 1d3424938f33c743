// The circuit half of a fawkes `Parameters` file: brotli blob -> Borsh gate stream -> CSR with dictionary-coded
// coefficients, streamed (a few MiB of decoded bytes exist at any time) and native -- the 1024-transaction system is
// 9.6e8 terms, the reference's own rollup more; a per-term interpreter loop is not an option.
//
// Replaces, at key-load time, what fawkes does again for EVERY proof:
//   WitnessCS::get_gate_iterator   /root/reference/fawkes-crypto/src/circuit/r1cs/cs.rs:243-245   brotli::Decompressor over Parameters.2
//   GateStreamedIterator::next     cs.rs:215-223   three parts per gate
//   read_gate_part                 cs.rs:193-213   u32 LE count | count x (32 B canonical LE Fr | u8 tag 0 = Input, 1 = Aux | u32 LE index)
//   the writer                     backend/bellman_groth16/setup.rs:25-32   CompressorWriter(_, 4096, quality 9, lgwin 22)
// Coefficients arrive canonical and leave in Montgomery form (ff-uint_derive/src/lib.rs:696-701: values >= r are InvalidData);
// variables become Input(i) -> i, Aux(j) -> num_input + j (cs.rs:255-268).
//
// Brotli itself (RFC 7932; the reference uses the `brotli` crate) is decoded by the system's libbrotlidec.so.1 -- present in
// this image and on the GPU box -- bound at run time with dlopen: the format's 122 KB static dictionary makes a private
// decoder pointless.  Without the library FK_GATES_BROTLI fails loudly with FK_ERR_UNSUPPORTED; FK_GATES_RAW needs nothing.
#include "common.hpp"
#include <new>
#include <stdexcept>
#include <algorithm>
#include <dlfcn.h>
#include <string.h>
#include <string>
#include <unordered_map>

struct fk_gates {
    uint32_t num_input = 0, num_aux = 0;
    uint64_t num_gates = 0;
    std::vector<uint64_t> ptr[3];
    std::vector<uint32_t> col[3], cidx[3];
    std::vector<fk::Fr> table;          // slot 0 = ONE
    uint64_t decoded_bytes = 0;
};

namespace fk {

// ------------------------------------------------------------------------------------------ libbrotlidec binding
struct BrotliApi {
    void *(*create)(void *, void *, void *) = nullptr;
    int (*stream)(void *, size_t *, const uint8_t **, size_t *, uint8_t **, size_t *) = nullptr;
    void (*destroy)(void *) = nullptr;
    bool ok = false;
    BrotliApi() {
        void *h = nullptr;
        for (const char *nm : {"libbrotlidec.so.1", "libbrotlidec.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        create = (decltype(create))dlsym(h, "BrotliDecoderCreateInstance");
        stream = (decltype(stream))dlsym(h, "BrotliDecoderDecompressStream");
        destroy = (decltype(destroy))dlsym(h, "BrotliDecoderDestroyInstance");
        ok = create && stream && destroy;
    }
};
static const BrotliApi &brotli_api() { static BrotliApi a; return a; }

// ------------------------------------------------------------------------------------------ incremental gate-stream parser
struct GateParser {
    fk_gates *g;
    uint64_t gate = 0; int part = 0; uint32_t left = 0; bool need_count = true;
    uint8_t carry[40]; size_t n_carry = 0;
    std::unordered_map<std::string, uint32_t> dict;
    uint8_t last_raw[32]; uint32_t last_idx = 0; bool have_last = false;
    std::string err;

    explicit GateParser(fk_gates *g_) : g(g_) {
        const Fr one = Fr::one();
        g->table.push_back(one);
        uint8_t c1[32] = {1};
        dict.emplace(std::string((const char *)c1, 32), 0u);
        // num_gates comes straight from a file header: reserve what a plausible stream needs, let push_back grow beyond it
        for (int k = 0; k < 3; k++) { g->ptr[k].reserve(std::min<uint64_t>(g->num_gates, (uint64_t)1 << 24) + 1); g->ptr[k].push_back(0); }
    }
    bool done() const { return gate == g->num_gates; }

    bool item(const uint8_t *p) {
        const uint8_t tag = p[32];
        uint32_t idx; memcpy(&idx, p + 33, 4);
        uint32_t v;
        if (tag == 0) { if (idx >= g->num_input) { err = "input index out of range"; return false; } v = idx; }
        else if (tag == 1) { if (idx >= g->num_aux) { err = "aux index out of range"; return false; } v = g->num_input + idx; }
        else { err = "enum elements overflow"; return false; }                  // cs.rs:209
        uint32_t ci;
        if (have_last && memcmp(p, last_raw, 32) == 0) ci = last_idx;
        else {
            std::string key((const char *)p, 32);
            auto it = dict.find(key);
            if (it == dict.end()) {
                Fr c; memcpy(&c, p, 32);
                for (int i = 7; i >= 0; i--) {           // canonical: strictly below r
                    if (c.v[i] < FrParams::p(i)) break;
                    if (c.v[i] > FrParams::p(i) || i == 0) { err = "non-canonical field element"; return false; }
                }
                if (g->table.size() >= 0xffffffffull) { err = "too many distinct coefficients"; return false; }
                it = dict.emplace(key, (uint32_t)g->table.size()).first;
                g->table.push_back(Fr::to_mont(c));
            }
            ci = it->second; memcpy(last_raw, p, 32); last_idx = ci; have_last = true;
        }
        g->col[part].push_back(v); g->cidx[part].push_back(ci);
        return true;
    }
    void close_parts() {       // empty linear combinations close at once
        while (!need_count && left == 0 && !done()) {
            g->ptr[part].push_back(g->col[part].size());
            need_count = true;
            if (++part == 3) { part = 0; gate++; }
        }
    }
    // consumes a chunk of decoded bytes; false on malformed data
    bool feed(const uint8_t *p, size_t n) {
        g->decoded_bytes += n;
        while (n) {
            if (done()) { err = "trailing bytes after the last gate"; return false; }
            const size_t want = need_count ? 4 : 37;
            if (n_carry || n < want) {                    // a record split across chunks: finish it in the carry buffer
                const size_t take = want - n_carry < n ? want - n_carry : n;
                memcpy(carry + n_carry, p, take); n_carry += take; p += take; n -= take;
                if (n_carry < want) return true;
                if (!unit(carry)) return false;
                n_carry = 0;
                continue;
            }
            if (!unit(p)) return false;
            p += want; n -= want;
        }
        return true;
    }
    bool unit(const uint8_t *p) {
        if (need_count) { memcpy(&left, p, 4); need_count = false; }
        else { if (!item(p)) return false; left--; }
        close_parts();
        return true;
    }
};

static int gates_decode(fk_ctx *ctx, const uint8_t *blob, size_t len, int format, uint32_t num_gates, uint32_t num_input, uint32_t num_aux, fk_gates **out) {
    if (!out || (len && !blob)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: null argument");
    *out = nullptr;
    if (num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: num_input must include the constant ONE");
    if ((uint64_t)num_input + num_aux > 0xffffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: too many variables");
    // every gate is at least three 4-byte counts: a header that promises more gates than the stream can hold is malformed
    if (format == FK_GATES_RAW && (uint64_t)num_gates * 12 > len) FK_SET_ERR(ctx, FK_ERR_FORMAT, "gates: %u gates cannot fit into %zu bytes", num_gates, len);
    // owns the half-built system and the decoder state on every way out, exceptions (std::bad_alloc) included
    struct Guard { fk_gates *g = nullptr; void *st = nullptr; const BrotliApi *br = nullptr; ~Guard() { if (st && br) br->destroy(st); delete g; } } guard;
    fk_gates *g = guard.g = new fk_gates();
    g->num_input = num_input; g->num_aux = num_aux; g->num_gates = num_gates;
    GateParser ps(g);
    auto fail = [&](int code, const std::string &msg) { ctx->err = "gates: " + msg; return code; };
    if (format == FK_GATES_RAW) {
        if (!ps.feed(blob, len)) return fail(FK_ERR_FORMAT, ps.err);
    } else if (format == FK_GATES_BROTLI) {
        const BrotliApi &br = brotli_api();
        if (!br.ok) return fail(FK_ERR_UNSUPPORTED, "libbrotlidec.so.1 not found (needed for a brotli gate blob)");
        void *st = br.create(nullptr, nullptr, nullptr);
        if (!st) return fail(FK_ERR_OOM, "brotli decoder allocation failed");
        guard.st = st; guard.br = &br;
        std::vector<uint8_t> buf((size_t)4 << 20);
        size_t avail_in = len; const uint8_t *next_in = blob;
        int res;
        do {
            size_t avail_out = buf.size(); uint8_t *next_out = buf.data();
            res = br.stream(st, &avail_in, &next_in, &avail_out, &next_out, nullptr);     // 0 error, 1 done, 2 needs input, 3 needs output
            if (res == 0) return fail(FK_ERR_FORMAT, "corrupt brotli stream");
            if (!ps.feed(buf.data(), buf.size() - avail_out)) return fail(FK_ERR_FORMAT, ps.err);
            if (res == 2 && avail_in == 0) return fail(FK_ERR_FORMAT, "brotli stream truncated");
        } while (res != 1);
    } else {
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "gates: unknown blob format %d", format);
    }
    if (ps.n_carry || !ps.done()) return fail(FK_ERR_FORMAT, "gate stream truncated (fewer than num_gates gates)");
    *out = g;
    guard.g = nullptr;
    return FK_OK;
}

}  // namespace fk

using namespace fk;

extern "C" {

int fk_gates_decode(fk_ctx *ctx, const uint8_t *blob, size_t len, int format, uint32_t num_gates, uint32_t num_input, uint32_t num_aux, fk_gates **out) { return fk_guard(ctx, [&]() -> int {
    fk_ctx local;                  // host-only routine: usable without a GPU context
    if (!ctx) ctx = &local;
    // a tiny brotli blob can inflate to anything: running out of host memory is a status code, never an exception that leaves
    // an extern "C" function (std::terminate would take the ctypes / Rust host down)
    try { return gates_decode(ctx, blob, len, format, num_gates, num_input, num_aux, out); }
    catch (const std::bad_alloc &) { if (out) *out = nullptr; ctx->err = "gates: out of host memory while decoding the gate stream"; return FK_ERR_OOM; }
    catch (const std::length_error &) { if (out) *out = nullptr; ctx->err = "gates: the gate stream is larger than this host can hold"; return FK_ERR_OOM; }
}); }

void fk_gates_free(fk_gates *g) { delete g; }

int fk_gates_info(const fk_gates *g, uint64_t out[8]) {
    if (!g || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[8] = {g->num_gates, g->col[0].size(), g->col[1].size(), g->col[2].size(), g->table.size(), g->decoded_bytes, g->num_input, g->num_aux};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

// one matrix as the arrays of an fk_r1cs: ptr[num_gates + 1], col[nnz], val[nnz x 4] (Montgomery; may be NULL)
int fk_gates_export(const fk_gates *g, int mtx, uint64_t *ptr, uint32_t *col, uint64_t *val) {
    if (!g || mtx < 0 || mtx > 2 || !ptr) return FK_ERR_BAD_ARG;
    memcpy(ptr, g->ptr[mtx].data(), g->ptr[mtx].size() * 8);
    const size_t nnz = g->col[mtx].size();
    if (col && nnz) memcpy(col, g->col[mtx].data(), nnz * 4);
    if (val) for (size_t i = 0; i < nnz; i++) memcpy(val + 4 * i, &g->table[g->cidx[mtx][i]], 32);
    return FK_OK;
}

}  // extern "C"

// the resident constraint system straight from the decoded stream (spmv.hip); its dictionary is reused as is
namespace fk { int r1cs_load_coded(fk_ctx *ctx, uint32_t num_input, uint32_t num_aux, uint64_t num_gates, const uint64_t *const ptr[3], const uint32_t *const col[3],
                                   const uint32_t *const cidx[3], const Fr *table, uint64_t n_table, fk_r1cs_dev **out); }

extern "C" int fk_r1cs_load_gates(fk_ctx *ctx, const fk_gates *g, fk_r1cs_dev **out) {
    if (!ctx || !g || !out) return FK_ERR_BAD_ARG;
    const uint64_t *ptr[3] = {g->ptr[0].data(), g->ptr[1].data(), g->ptr[2].data()};
    const uint32_t *col[3] = {g->col[0].data(), g->col[1].data(), g->col[2].data()};
    const uint32_t *cidx[3] = {g->cidx[0].data(), g->cidx[1].data(), g->cidx[2].data()};
    try { return r1cs_load_coded(ctx, g->num_input, g->num_aux, g->num_gates, ptr, col, cidx, g->table.data(), g->table.size(), out); }
    catch (const std::bad_alloc &) { *out = nullptr; ctx->err = "r1cs: out of host memory"; return FK_ERR_OOM; }
}
