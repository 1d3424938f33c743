// Device-resident R1CS and the sparse evaluation  a = A z, b = B z, c = C z  (SURVEY.md section 8f, row 1).
//
// Replaces, for the prover, what bellman's ProvingAssignment does gate by gate on the CPU while
// fawkes streams the constraint system out of its brotli blob
// (/root/reference/fawkes-crypto/src/backend/bellman_groth16/mod.rs:92-99,
//  /root/reference/fawkes-crypto/src/circuit/r1cs/cs.rs:184-223; evaluation semantics SURVEY App. A.1):
// a_i = <A_i, z>, b_i = <B_i, z>, c_i = <C_i, z>, plus one `input_i * 0 = 0` row per input, and the
// three density maps, which are structural (a variable is marked when it is *visited*) and are therefore
// computed once at load time.
//
// Layout in HBM: CSR per matrix -- row_ptr u64[num_gates+1], col u32[nnz] (Input(i) -> i,
// Aux(j) -> num_input + j) and a coefficient DICTIONARY: fawkes LCs carry few distinct constants
// (1, -1, small integers, Poseidon MDS entries), so each term stores a u32 index into a table of unique
// 32-byte Montgomery values instead of the value itself: 8 B per term instead of 36 B, and the table
// stays in L2.  Index 0 is reserved for ONE (multiply skipped, exactly as bellman's eval does).
// Kernel: one lane per (matrix, row); HBM-bound gather of z (32 B per term).
#include "common.hpp"
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <string>

#include "r1cs.hpp"

namespace fk {

struct SpmvArgs {
    const uint64_t *ptr[3];
    const uint32_t *col[3];
    const uint32_t *cidx[3];
    Fr *out[3];
};

// G = 2^lg lanes share one row: lane s takes terms s, s + G, ... and the partial sums are folded with wave64 shuffles.
// Real circuits have long linear combinations (the eddsa verifier: 133 terms per gate on average, rows of up to 512), and
// with one lane per row a wave runs as long as its longest row; the synthetic rollup shape (1-2 terms per row) keeps G = 1.
//
// TILED: the CSR describes one instance of a batch circuit and row t is row t % base_gates of copy t / base_gates; the
// copy's variables are found by arithmetic (ONE shared, then every copy's inputs, then every copy's aux variables: the
// layout of a circuit that allocates the same gadget `copies` times), so a 4096-signature batch costs the memory of one.
struct TileDims { uint32_t base_gates, base_input, base_aux; };
// binned: bit k set = matrix k's gate rows belong to spmv_binned_kernel; this kernel then only writes the rows behind them
// SLICED: lane group tl evaluates row t = rank + W * tl and writes out[tl] (the cyclic slice of rank `rank` of W = 2^log_w)
template <bool TILED, bool SLICED>
__global__ __launch_bounds__(256) void spmv_kernel(SpmvArgs a, const Fr *table, const Fr *z, uint64_t num_gates, uint32_t num_input, uint32_t lg0, uint32_t lg1,
                                                   uint32_t lg2, TileDims td, uint32_t binned, uint32_t log_w, uint32_t rank) {
    const uint32_t mtx = blockIdx.y;
    const uint32_t lg = mtx == 0 ? lg0 : (mtx == 1 ? lg1 : lg2);
    const uint64_t tl = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lg;
    const uint64_t t = SLICED ? rank + (tl << log_w) : tl;
    const uint32_t sub = threadIdx.x & ((1u << lg) - 1);
    const uint64_t rows = num_gates + num_input;
    if (t >= rows) return;                   // the lanes of a row group leave together (256 is a multiple of G)
    if (((binned >> mtx) & 1) && t < num_gates) return;
    const uint64_t *ptr = a.ptr[mtx]; const uint32_t *col = a.col[mtx]; const uint32_t *cidx = a.cidx[mtx];
    Fr acc = Fr::zero();
    if (t < num_gates) {
        uint32_t copy = 0; uint64_t row = t;
        if (TILED) { copy = (uint32_t)t / td.base_gates; row = (uint32_t)t - copy * td.base_gates; }
        const uint32_t in_off = copy * (td.base_input - 1), aux_off = num_input + copy * td.base_aux - td.base_input;
        for (uint64_t k = ptr[row] + sub, e = ptr[row + 1]; k < e; k += (uint64_t)1 << lg) {
            uint32_t cv = col[k];
            if (TILED && cv) cv += cv < td.base_input ? in_off : aux_off;
            Fr v = z[cv];
            const uint32_t ci = cidx[k];
            if (ci) v = Fr::mul(v, table[ci]);
            acc = Fr::add(acc, v);
        }
    } else if (mtx == 0 && sub == 0) {
        acc = z[t - num_gates];        // bellman's extra rows: input_i * 0 = 0
    }
    for (uint32_t off = (1u << lg) >> 1; off; off >>= 1) {
        Fr o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_down((int)acc.v[i], off, 64);
        acc = Fr::add(acc, o);
    }
    if (sub == 0) a.out[mtx][tl] = acc;
}

// Circuits built from Poseidon are bimodal: three quarters of the eddsa verifier's rows hold ONE term, the rest 128-512 (the
// MDS mixing is kept as lazily expanded linear combinations).  A wave runs as long as its longest row, so one lane-group
// size for a whole matrix wastes most lanes either way.  Here the rows of a matrix are split by length into classes with
// their own group size (1, 2, 4, 8 or 16 lanes per row: a lane gets 4 .. 7 terms) and sorted by length inside a class, so that the rows sharing a wave
// have (nearly) the same length; long rows take two terms per step through the dual-chain multiplier.  Segment s of the
// launch is one (matrix, class); a tiled system walks its copies in the outer order, so a copy's z stays in cache.
// SLICED (b.rowlist = the residue-grouped lists, b.first_block = this rank's plan): group index -> (copy, position in the
// residue group that copy needs); the row's result goes to out[(copy * base_gates + row) >> log_w].
// MODE 1 (experiment builds, FK_SPMV_FAKE4=1, TIMING ONLY -- the results are wrong): the kernel streams 4 bytes per term instead of 8 -- the coefficient
// index is derived from the column (no cidx load) -- which is what a packed form (u16 coefficient index + u16 column offset inside a row block's
// window) would stream at best: an upper bound on what that form can save (tools/spmv_untiled_probe.py).
#ifndef FK_SPMV_MINB
#define FK_SPMV_MINB 1      // workgroups per compute unit the register allocation is held to (256 lanes = one wave per SIMD each): see the kernel's comment
#endif
template <bool TILED, bool SLICED, int MODE = 0>
__global__ __launch_bounds__(256, FK_SPMV_MINB) void spmv_binned_kernel(SpmvArgs a, BinArgs b, const Fr *table, const Fr *z, uint32_t num_input, TileDims td, uint32_t copies, SliceArgs sl) {
    uint32_t s = 0;
    while (s + 1 < b.nseg && blockIdx.x >= b.first_block[s + 1]) s++;
    const uint32_t lg = b.lg[s], mtx = b.mtx[s], nr = b.n_rows[s];
    const uint64_t gidx = (uint64_t)(blockIdx.x - b.first_block[s]) * (256u >> lg) + (threadIdx.x >> lg);
    const uint32_t sub = threadIdx.x & ((1u << lg) - 1);
    uint32_t copy = 0, li = (uint32_t)gidx;
    if (SLICED) {
        const uint32_t T = sl.T[s];
        const uint64_t q = gidx / T;
        uint32_t rem = (uint32_t)(gidx - q * T), pp = 0;
        while (pp + 1 < sl.P && rem >= sl.cnt[s][sl.rho[pp]]) { rem -= sl.cnt[s][sl.rho[pp]]; pp++; }
        const uint64_t cp = q * sl.P + pp;
        if (cp >= copies || rem >= sl.cnt[s][sl.rho[pp]]) return;
        copy = (uint32_t)cp;
        li = sl.offs[s][sl.rho[pp]] + rem;
    } else {
        if (gidx >= (uint64_t)nr * copies) return;
        if (TILED) { copy = (uint32_t)(gidx / nr); li = (uint32_t)(gidx - (uint64_t)copy * nr); }
    }
    const uint32_t row = b.rowlist[mtx][b.list_off[s] + li];
    const uint32_t in_off = copy * (td.base_input - 1), aux_off = num_input + copy * td.base_aux - td.base_input;
    const uint64_t *ptr = a.ptr[mtx]; const uint32_t *col = a.col[mtx]; const uint32_t *cidx = a.cidx[mtx];
    // MODE 2 (experiment builds, FK_SPMV_SEQ=1, TIMING ONLY): the row's terms are read from where a layout PERMUTED into class-list order would
    // hold them -- [pptr[i], pptr[i + 1]) at list position i: neighbouring lane groups read neighbouring pointers and neighbouring runs of col / cidx,
    // no rowlist -> ptr -> col chain (the row id is only needed for the store) -- same row lengths, the terms themselves are another row's
    const uint64_t pi = (uint64_t)b.list_off[s] + li;
    const uint64_t e = MODE == 2 ? b.pptr[mtx][pi + 1] : ptr[row + 1];
    uint64_t k = (MODE == 2 ? b.pptr[mtx][pi] : ptr[row]) + sub;
    const uint32_t G = 1u << lg;
    auto var = [&](uint32_t cv) -> uint32_t { if (TILED && cv) cv += cv < td.base_input ? in_off : aux_off; return cv; };
    Fr acc = Fr::zero();
    // four terms per step with ONE Montgomery reduction (Fp::dot4: 328 multiply-accumulates instead of 544) ...
    for (; k + 3 * (uint64_t)G < e; k += 4 * G) {
        const uint32_t c0 = var(col[k]), c1 = var(col[k + G]), c2 = var(col[k + 2 * G]), c3 = var(col[k + 3 * G]);
        const uint32_t i0 = (MODE == 1) ? (c0 & 4095u) : cidx[k], i1 = (MODE == 1) ? (c1 & 4095u) : cidx[k + G], i2 = (MODE == 1) ? (c2 & 4095u) : cidx[k + 2 * G], i3 = (MODE == 1) ? (c3 & 4095u) : cidx[k + 3 * G];
        acc = Fr::add(acc, Fr::dot4(z[c0], table[i0], z[c1], table[i1], z[c2], table[i2], z[c3], table[i3]));
    }           // (loading the NEXT step's indices ahead of this step's products was measured: 169.8 against 168.8 ms per proof)
    if (lg) {   // ... then two (one lane per row: the last terms one by one, ONE coefficients skipped)
        Fr acc1 = Fr::zero();
        for (; k + G < e; k += 2 * G) {        // table[0] is ONE: multiplying by it returns the (reduced) value itself
            const uint32_t c0 = var(col[k]), c1 = var(col[k + G]), i0 = (MODE == 1) ? (c0 & 4095u) : cidx[k], i1 = (MODE == 1) ? (c1 & 4095u) : cidx[k + G];
            Fr p0, p1;
            Fr::mul2(z[c0], table[i0], z[c1], table[i1], p0, p1);
            Fr::add2(acc, p0, acc1, p1, acc, acc1);
        }
        acc = Fr::add(acc, acc1);
    }
    for (; k < e; k += G) {
        const uint32_t cv = var(col[k]);
        Fr v = z[cv];
        const uint32_t ci = (MODE == 1) ? (cv & 4095u) : cidx[k];
        if (ci) v = Fr::mul(v, table[ci]);
        acc = Fr::add(acc, v);
    }
    for (uint32_t off = G >> 1; off; off >>= 1) {
        Fr o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_down((int)acc.v[i], off, 64);
        acc = Fr::add(acc, o);
    }
    if (sub == 0) a.out[mtx][((uint64_t)copy * td.base_gates + row) >> (SLICED ? sl.log_w : 0)] = acc;
}

// Batch circuits (TILED), rows of middling length the other way round: one WAVE per (instance row, block of 64 copies) -- lane l
// evaluates the row for copy 64 * cb + l.  Every lane of a wave then walks the SAME row: the trip count, the column and
// coefficient indices and the coefficients are wave-uniform (scalar loads, one copy in SGPRs), no lane is padding (a 33-term
// row shared by 16 lanes keeps them busy 33 / 48 of the time), there are no partial sums to fold, and four terms at a time
// share one Montgomery reduction.  What differs per lane is the copy's offset into z.  Measured on the benchmark's system
// (tools/spmv_rollup_probe.py, profiles/r03_spmv_rollup_probe.log): rows of 4 .. 31 terms 2.16 -> 1.78 ms, 32 .. 63 terms
// 4.58 -> 2.83 ms; rows of 64 terms and more are faster in the lane-group form (their 16 lanes read neighbouring variables of
// ONE copy; here 64 lanes read 64 copies' -- 3.72 -> 4.80 ms), single-term rows too, so those stay with spmv_binned_kernel.
// wavelist: the rows (matrix in the top two bits) in circuit order, so that what runs at one time reads a window of the
// same 64 copies' variables; copy block outermost.
__global__ __launch_bounds__(256) void spmv_tiled_wave_kernel(SpmvArgs a, const Fr *table, const Fr *z, uint32_t num_input, TileDims td, uint32_t copies,
                                                              const uint32_t *wavelist, uint32_t n_list, uint32_t n_waves) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (wave >= n_waves) return;
    const uint32_t cb = wave / n_list, ent = wavelist[wave - cb * n_list], mtx = ent >> 30, row = ent & 0x3fffffffu;
    const uint32_t c_raw = cb * 64 + lane;
    const bool active = c_raw < copies;
    const uint32_t copy = active ? c_raw : copies - 1;
    const uint32_t in_off = copy * (td.base_input - 1), aux_off = num_input + copy * td.base_aux - td.base_input;
    const uint64_t *ptr = a.ptr[mtx]; const uint32_t *col = a.col[mtx]; const uint32_t *cidx = a.cidx[mtx];
    auto var = [&](uint32_t cv) -> uint32_t { if (cv) cv += cv < td.base_input ? in_off : aux_off; return cv; };
    uint64_t k = ptr[row];
    const uint64_t e = ptr[row + 1];
    Fr acc = Fr::zero();
    for (; k + 3 < e; k += 4)
        acc = Fr::add(acc, Fr::dot4(z[var(col[k])], table[cidx[k]], z[var(col[k + 1])], table[cidx[k + 1]], z[var(col[k + 2])], table[cidx[k + 2]],
                                    z[var(col[k + 3])], table[cidx[k + 3]]));
    if (k + 1 < e) {
        Fr p0, p1;
        Fr::mul2(z[var(col[k])], table[cidx[k]], z[var(col[k + 1])], table[cidx[k + 1]], p0, p1);
        acc = Fr::add(acc, Fr::add(p0, p1));
        k += 2;
    }
    if (k < e) {
        Fr v = z[var(col[k])];
        const uint32_t ci = cidx[k];
        if (ci) v = Fr::mul(v, table[ci]);       // index 0 is ONE: bellman's eval skips that product, and so does every lane here
        acc = Fr::add(acc, v);
    }
    if (active) a.out[mtx][(uint64_t)copy * td.base_gates + row] = acc;
}

// the largest variable index each row window reads: terms [tb[j], tb[j + 1]) of a matrix belong to window j
struct WinBounds { uint64_t tb[fk_r1cs_dev::WIN_MAX + 1]; uint32_t k; };
__global__ __launch_bounds__(256) void col_window_max_kernel(const uint32_t *col, WinBounds wb, uint32_t *out) {
    const uint64_t n = wb.tb[wb.k];
    uint32_t w = 0, mx = 0; bool any = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        while (i >= wb.tb[w + 1]) { if (any) atomicMax(&out[w], mx); mx = 0; any = false; w++; }
        const uint32_t c = col[i];
        if (c > mx) mx = c;
        any = true;
    }
    if (any) atomicMax(&out[w], mx);
}

}  // namespace fk

using namespace fk;

extern "C" {

void fk_r1cs_free(fk_ctx *ctx, fk_r1cs_dev *r) {
    if (!r) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (int k = 0; k < 3; k++) { if (r->ptr[k]) (void)hipFree(r->ptr[k]); if (r->col[k]) (void)hipFree(r->col[k]); if (r->cidx[k]) (void)hipFree(r->cidx[k]); if (r->rowlist[k]) (void)hipFree(r->rowlist[k]); if (r->pptr[k]) (void)hipFree(r->pptr[k]); }
    for (void *p : {(void *)r->table, (void *)r->d_a_aux, (void *)r->d_b_in, (void *)r->d_b_aux, (void *)r->d_idx_a, (void *)r->d_idx_b}) if (p) (void)hipFree(p);
    for (auto &sl : r->slices) for (uint32_t *p : sl.d_list) if (p) (void)hipFree(p);
    if (r->d_wavelist) (void)hipFree(r->d_wavelist);
    delete r;
}

// cs: one instance; the resident system stands for `copies` of it (1 = the system itself).  pre_cidx / pre_table: the
// coefficients already dictionary-coded (gatestream.hip: slot 0 = ONE), cs->*_val then unused.
// pre_density: a_aux / b_in / b_aux flags the caller already derived (the gate decoder does, while it parses): the arrays then come
// from this library's own decoder, which has range-checked every index -- the per-term passes over the host arrays are skipped.
static int r1cs_load_impl(fk_ctx *ctx, const fk_r1cs *cs, uint32_t copies, fk_r1cs_dev **out, const uint32_t *const *pre_cidx = nullptr,
                          const Fr *pre_table = nullptr, uint64_t n_pre_table = 0, const uint8_t *const *pre_density = nullptr) {
    if (!ctx || !cs || !out) return FK_ERR_BAD_ARG;
    *out = nullptr;
    if (copies == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: copies must be at least 1");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t *ptrs[3] = {cs->a_ptr, cs->b_ptr, cs->c_ptr};
    const uint32_t *cols[3] = {cs->a_col, cs->b_col, cs->c_col};
    const uint64_t *vals[3] = {cs->a_val, cs->b_val, cs->c_val};
    const uint64_t nv = (uint64_t)cs->num_input + cs->num_aux;
    if (cs->num_input == 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: num_input must include the constant ONE");
    for (int k = 0; k < 3; k++) {
        if (!ptrs[k]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: null row pointer");
        if (ptrs[k][0] != 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: row_ptr[0] must be 0");
        if (pre_density && pre_cidx) continue;
        for (uint64_t g = 0; g < cs->num_gates; g++) if (ptrs[k][g + 1] < ptrs[k][g]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: row_ptr not monotone");
        const uint64_t nnz = ptrs[k][cs->num_gates];
        if (nnz && !cols[k]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: null column array");   // vals[k] == NULL: all coefficients ONE
        for (uint64_t i = 0; i < nnz; i++) if (cols[k][i] >= nv) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: variable index %u out of range", cols[k][i]);
    }
    // totals of the tiled system: ONE is shared, inputs and aux variables are per copy
    const uint64_t t_in = 1 + (uint64_t)copies * (cs->num_input - 1), t_aux = (uint64_t)copies * cs->num_aux, t_gates = (uint64_t)copies * cs->num_gates;
    if (t_in + t_aux > 0xffffffffull || t_gates + t_in > 0xffffffffull || (copies > 1 && cs->num_gates == 0))
        FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: %u copies of this system do not fit 32-bit variable / row indices", copies);
    const auto t_load0 = std::chrono::steady_clock::now();
    double t_copy = 0;
    fk_r1cs_dev *r = new fk_r1cs_dev();
    r->num_input = (uint32_t)t_in; r->num_aux = (uint32_t)t_aux; r->num_gates = t_gates;
    r->copies = copies; r->base_input = cs->num_input; r->base_aux = cs->num_aux; r->base_gates = (uint32_t)cs->num_gates;
    // coefficient dictionary; slot 0 = ONE
    std::unordered_map<std::string, uint32_t> dict;
    std::vector<Fr> table;
    const Fr one = Fr::one();
    table.push_back(one);
    dict.emplace(std::string((const char *)&one, 32), 0u);
    if (pre_cidx) {
        if (!pre_table || !n_pre_table || pre_table[0] != one) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: coefficient table must start with ONE");
        table.assign(pre_table, pre_table + n_pre_table);
    }
    std::vector<uint8_t> a_aux(cs->num_aux ? cs->num_aux : 1, 0), b_in(cs->num_input, 0), b_aux(cs->num_aux ? cs->num_aux : 1, 0);
    const bool trusted = pre_density && pre_cidx;
    if (trusted) {
        if (cs->num_aux) { memcpy(a_aux.data(), pre_density[0], cs->num_aux); memcpy(b_aux.data(), pre_density[2], cs->num_aux); }
        memcpy(b_in.data(), pre_density[1], cs->num_input);
    }
    int rc = FK_OK;
    auto fail = [&](int code) { fk_r1cs_free(ctx, r); return code; };
    // rows of [wave_lo, wave_hi) terms of a batch of >= wave_copies copies go to spmv_tiled_wave_kernel (see there).  The length
    // classes are 16 lanes per row from 64 terms, 8 from 32, 4 from 16, 2 from 8, one lane below; with [4, 64) the wave kernel takes
    // the WHOLE 8-, 4- and 2-lane classes and the head (4 .. 7 terms) of the one-lane class.  The clipping in r1cs_eval_impl removes a
    // prefix or a suffix of a class segment, so the range may cut only the first class (wave_hi >= 64) and the last one (wave_lo < 8):
    // an upper bound strictly inside a middle class is snapped down to that class's boundary (experiment builds can set any value).
    static const uint32_t wave_copies = (uint32_t)tune("FK_SPMV_WAVE_MIN_COPIES", 64), wave_lo = 4;
    static const uint32_t wave_hi = [] {
        const uint32_t v = (uint32_t)std::min(1024, std::max(32, tune("FK_SPMV_WAVE_HI", 64)));
        return v >= 64 ? v : 32u;           // 32 .. 63 -> 32: the boundary between the 8-lane and the 4-lane class
    }();
    uint64_t bin_min = 8;          // rows this long make a matrix binned; FK_SPMV_BIN_MIN=0 turns the binned product off (a run-time switch: the tests run both kernels)
    if (const char *e = getenv("FK_SPMV_BIN_MIN")) { bin_min = strtoull(e, nullptr, 10); if (!bin_min) bin_min = ~0ull; }
    // The class lists of a matrix depend on its row lengths alone: the three are built side by side (a third of a second each for the
    // benchmark's 33.5 M rows) before the matrices are uploaded.
    struct ListPlan { bool binned = false; std::vector<uint32_t> list; uint32_t cls_n[SPMV_CLASSES] = {0}, wave_from = 0, wave_to = 0; };
    ListPlan plans[3];
    static const uint32_t cls_lo[SPMV_CLASSES] = {64, 32, 16, 8, 0}, cls_lg[SPMV_CLASSES] = {4, 3, 2, 1, 0};
    auto build_lists = [&](int k) {
        ListPlan &lp = plans[k];
        uint64_t maxlen = 0;
        for (uint64_t g = 0; g < cs->num_gates; g++) if (ptrs[k][g + 1] - ptrs[k][g] > maxlen) maxlen = ptrs[k][g + 1] - ptrs[k][g];
        if (!(maxlen >= bin_min && cs->num_gates && cs->num_gates < 0xffffffffull)) return;
        lp.binned = true;
        const uint32_t ng = (uint32_t)cs->num_gates, CAP = 1024;            // counting sort by min(length, CAP), descending, stable
        std::vector<uint32_t> cnt(CAP + 2, 0);
        std::vector<uint32_t> &list = lp.list;
        list.resize(ng);
        auto key = [&](uint32_t g) { const uint64_t l = ptrs[k][g + 1] - ptrs[k][g]; return (uint32_t)(l < CAP ? l : CAP); };
        for (uint32_t g = 0; g < ng; g++) cnt[CAP - key(g) + 1]++;
        for (uint32_t i = 0; i <= CAP; i++) cnt[i + 1] += cnt[i];
        for (uint32_t g = 0; g < ng; g++) list[cnt[CAP - key(g)]++] = g;
        // classes: 16 lanes per row from 64 terms, 8 from 32, 4 from 16, 2 from 8, one lane below -- a lane then has 4 .. 7 terms
        // (one or two four-term steps with a shared reduction) in every class but the last; cnt[i] is now the END of key CAP - i
        uint32_t cls_start[SPMV_CLASSES];
        for (int c = 0; c < SPMV_CLASSES; c++) {
            cls_start[c] = c ? cnt[CAP - cls_lo[c - 1]] : 0;
            lp.cls_n[c] = (cls_lo[c] ? cnt[CAP - cls_lo[c]] : ng) - cls_start[c];
        }
        // A large flat system (a circuit as it comes out of a Parameters file): sorting a class by length over the WHOLE system
        // puts rows from everywhere in the circuit side by side -- every wave then gathers z from all over the witness and
        // streams its matrix entries from all over the CSR.  Sort by length inside blocks of consecutive rows instead: the rows
        // that share a wave still have (nearly) the same length, and what runs at one time reads one window of the circuit
        // (the benchmark's system with every term explicit: evaluation 15.2 -> see profiles/r03_spmv_rollup_probe.log).
        static const uint32_t block_rows = (uint32_t)std::max(0, tune("FK_SPMV_BLOCK_ROWS", 4096));
        if (copies == 1 && block_rows && ng >= 16 * block_rows) {
            uint32_t pos[SPMV_CLASSES];                            // class c holds the rows of cls_lo[c] <= length < cls_lo[c - 1]
            for (int c = 0; c < SPMV_CLASSES; c++) pos[c] = cls_start[c];
            std::vector<uint32_t> bc(CAP + 2);
            std::vector<uint32_t> tmp(block_rows);
            for (uint32_t g0 = 0; g0 < ng; g0 += block_rows) {
                const uint32_t g1 = std::min(ng, g0 + block_rows);
                std::fill(bc.begin(), bc.end(), 0u);
                for (uint32_t g = g0; g < g1; g++) bc[CAP - key(g) + 1]++;
                for (uint32_t i = 0; i <= CAP; i++) bc[i + 1] += bc[i];
                for (uint32_t g = g0; g < g1; g++) tmp[bc[CAP - key(g)]++] = g;           // the block's rows, longest first (stable)
                for (uint32_t i = 0; i < g1 - g0; i++) {
                    const uint32_t l = key(tmp[i]);
                    int c = 0;
                    while (l < cls_lo[c]) c++;
                    list[pos[c]++] = tmp[i];
                }
            }
        }
        lp.wave_from = cnt[CAP - wave_hi]; lp.wave_to = cnt[CAP - wave_lo];
    };
    if (cs->num_gates >= ((uint64_t)1 << 20)) {
        std::atomic<bool> threw{false};
        auto guarded = [&](int k) { try { build_lists(k); } catch (...) { threw = true; } };       // (an exception must not leave a thread)
        {
            std::thread t1([&] { guarded(1); });
            std::thread t2;
            try { t2 = std::thread([&] { guarded(2); }); } catch (...) { guarded(2); }
            guarded(0);
            t1.join(); if (t2.joinable()) t2.join();
        }
        if (threw) { ctx->err = "r1cs: out of host memory while ordering the rows"; return fail(FK_ERR_OOM); }
    } else for (int k = 0; k < 3; k++) build_lists(k);
    for (int k = 0; k < 3 && rc == FK_OK; k++) {
        const uint64_t nnz = ptrs[k][cs->num_gates];
        r->nnz[k] = nnz * copies;
        std::vector<uint32_t> cidx(trusted ? 1 : nnz ? nnz : 1);
        Fr last = one; uint32_t last_idx = 0;        // one-entry cache in front of the hash map
        for (uint64_t i = 0; i < nnz && !trusted; i++) {
            uint32_t ci = 0;
            if (pre_cidx) {
                ci = pre_cidx[k][i];
                if (ci >= n_pre_table) { ctx->err = "r1cs: coefficient index out of range"; return fail(FK_ERR_BAD_ARG); }
            } else if (vals[k] && memcmp(vals[k] + 4 * i, &one, 32) != 0) {
                if (memcmp(vals[k] + 4 * i, &last, 32) == 0) ci = last_idx;
                else {
                    std::string key((const char *)(vals[k] + 4 * i), 32);
                    auto it = dict.find(key);
                    if (it == dict.end()) {
                        if (table.size() >= 0xffffffffull) { ctx->err = "r1cs: too many distinct coefficients"; return fail(FK_ERR_BAD_ARG); }
                        Fr v; memcpy(&v, vals[k] + 4 * i, 32);
                        it = dict.emplace(key, (uint32_t)table.size()).first;
                        table.push_back(v);
                    }
                    ci = it->second; memcpy(&last, vals[k] + 4 * i, 32); last_idx = ci;
                }
            }
            cidx[i] = ci;
            const uint32_t v = cols[k][i];
            if (k == 0) { if (v >= cs->num_input) a_aux[v - cs->num_input] = 1; }
            else if (k == 1) { if (v < cs->num_input) b_in[v] = 1; else b_aux[v - cs->num_input] = 1; }
        }
        const auto tc0 = std::chrono::steady_clock::now();
        if (hipMalloc((void **)&r->ptr[k], (cs->num_gates + 1) * 8) != hipSuccess || hipMalloc((void **)&r->col[k], (nnz + 1) * 4) != hipSuccess ||
            hipMalloc((void **)&r->cidx[k], (nnz + 1) * 4) != hipSuccess) { ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM); }
        if (hipMemcpy(r->ptr[k], ptrs[k], (cs->num_gates + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
        if (nnz && hipMemcpy(r->col[k], cols[k], nnz * 4, hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
        if (nnz && hipMemcpy(r->cidx[k], trusted ? pre_cidx[k] : cidx.data(), nnz * 4, hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
        t_copy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
        // length classes (spmv_binned_kernel) when the matrix has long rows: planned above (build_lists)
        ListPlan &lp = plans[k];
        if (lp.binned && rc == FK_OK) {
            const uint32_t ng = (uint32_t)cs->num_gates;
            std::vector<uint32_t> &list = lp.list;
            const uint32_t *cls_n = lp.cls_n;
            r->wave_from[k] = lp.wave_from; r->wave_to[k] = lp.wave_to;             // rows of wave_lo <= length < wave_hi
            if (hipMalloc((void **)&r->rowlist[k], (size_t)ng * 4) != hipSuccess) { ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM); }
            if (hipMemcpy(r->rowlist[k], list.data(), (size_t)ng * 4, hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
#ifdef FK_EXPERIMENTS
            if (tune("FK_SPMV_SEQ", 0) && copies == 1 && rc == FK_OK) {      // pointers of the layout permuted into class-list order (timing experiment)
                std::vector<uint64_t> pp((size_t)ng + 1);
                pp[0] = 0;
                for (uint32_t i = 0; i < ng; i++) pp[i + 1] = pp[i] + (ptrs[k][list[i] + 1] - ptrs[k][list[i]]);
                if (hipMalloc((void **)&r->pptr[k], ((size_t)ng + 1) * 8) != hipSuccess) { ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM); }
                if (hipMemcpy(r->pptr[k], pp.data(), ((size_t)ng + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
                r->bins.pptr[k] = r->pptr[k];
            }
#endif
            r->h_rowlist[k] = std::move(list);
            BinArgs &b = r->bins;
            b.rowlist[k] = r->rowlist[k]; b.mask |= 1u << k;
            uint32_t off = 0;
            for (int c = 0; c < SPMV_CLASSES; c++) {
                if (cls_n[c]) {
                    const uint64_t groups = (uint64_t)cls_n[c] * copies, per = 256u >> cls_lg[c], blocks = (groups + per - 1) / per;
                    if (b.first_block[b.nseg] + blocks > 0x7fffffffull) { ctx->err = "r1cs: system too large for the binned product"; return fail(FK_ERR_BAD_ARG); }
                    b.lg[b.nseg] = cls_lg[c]; b.mtx[b.nseg] = k; b.n_rows[b.nseg] = cls_n[c]; b.list_off[b.nseg] = off;
                    b.first_block[b.nseg + 1] = b.first_block[b.nseg] + (uint32_t)blocks;
                    b.nseg++;
                }
                off += cls_n[c];
            }
        }
    }
    if (getenv("FK_GATES_TRACE")) fprintf(stderr, "[fk] r1cs load: %.2f s so far, %.2f of them allocating and copying the matrices\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load0).count(), t_copy);
    if (rc != FK_OK) { ctx->err = "r1cs: upload failed"; return fail(rc); }
    // row windows for the chunked hand-over of fk_prove_r1cs (r1cs.hpp): explicit systems with block-sorted class lists, every matrix binned
    {
        static const uint32_t block_rows_w = (uint32_t)std::max(0, tune("FK_SPMV_BLOCK_ROWS", 4096));
        static const int t_win = std::min<int>(fk_r1cs_dev::WIN_MAX, std::max(0, tune("FK_SPMV_WINDOWS", 8)));
        const uint64_t ng = cs->num_gates;
        if (copies == 1 && t_win >= 2 && block_rows_w && ng >= 16ull * block_rows_w && ng < 0xffffffffull && r->bins.mask == 7u) {
            const uint32_t K = (uint32_t)t_win;
            for (uint32_t j = 0; j <= K; j++) r->win_row[j] = j == K ? ng : (ng * j / K) / block_rows_w * block_rows_w;
            for (uint32_t s = 0; s < r->bins.nseg; s++) {
                const std::vector<uint32_t> &lst = r->h_rowlist[r->bins.mtx[s]];
                const uint32_t *lo = lst.data() + r->bins.list_off[s], *hi = lo + r->bins.n_rows[s];
                for (uint32_t j = 0; j <= K; j++) {
                    const uint64_t bound = r->win_row[j];
                    r->win_cnt[j][s] = (uint32_t)(std::partition_point(lo, hi, [&](uint32_t row) { return row < bound; }) - lo);       // a prefix: blocks of rows ascend inside a class
                }
            }
            uint32_t *d_mx = nullptr;
            if (hipMalloc((void **)&d_mx, K * 4) != hipSuccess || hipMemset(d_mx, 0, K * 4) != hipSuccess) { if (d_mx) (void)hipFree(d_mx); ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM); }
            for (int k = 0; k < 3; k++) {
                WinBounds wb{}; wb.k = K;
                for (uint32_t j = 0; j <= K; j++) wb.tb[j] = ptrs[k][r->win_row[j]];
                if (wb.tb[K]) hipLaunchKernelGGL(col_window_max_kernel, dim3(2048), dim3(256), 0, ctx->stream, r->col[k], wb, d_mx);
            }
            uint32_t mx[fk_r1cs_dev::WIN_MAX] = {0};
            const bool ok = hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(mx, d_mx, K * 4, hipMemcpyDeviceToHost) == hipSuccess;
            (void)hipFree(d_mx);
            if (!ok) { ctx->err = "r1cs: window planning failed"; return fail(FK_ERR_HIP); }
            uint64_t run = cs->num_input;                          // the rows behind the gates (input_i * 0 = 0) read the inputs: part of the first piece
            for (uint32_t j = 0; j < K; j++) { run = std::max<uint64_t>(run, (uint64_t)mx[j] + 1); r->win_need[j] = j + 1 == K ? nv : run; }
            r->win_k = K;
        }
    }
    if (copies > 1 && wave_copies && copies >= wave_copies && r->bins.mask && cs->num_gates < ((uint64_t)1 << 30)) {
        std::vector<uint32_t> wl;
        for (uint64_t g = 0; g < cs->num_gates; g++)
            for (uint32_t k = 0; k < 3; k++) {
                const uint64_t l = ptrs[k][g + 1] - ptrs[k][g];
                if (((r->bins.mask >> k) & 1) && l >= wave_lo && l < wave_hi) wl.push_back(k << 30 | (uint32_t)g);
            }
        if (!wl.empty() && (uint64_t)((copies + 63) / 64) * wl.size() < 0xfffffff0ull) {
            if (hipMalloc((void **)&r->d_wavelist, wl.size() * 4) != hipSuccess) { ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM); }
            if (hipMemcpy(r->d_wavelist, wl.data(), wl.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "r1cs: upload failed"; return fail(FK_ERR_HIP); }
            r->n_wavelist = (uint32_t)wl.size();
        }
    }
    if (copies > 1) {      // density maps of the whole batch: every copy repeats the instance's pattern, ONE is shared
        auto rep = [&](std::vector<uint8_t> &f, size_t skip, size_t per) {
            std::vector<uint8_t> o(skip + per * copies + (skip + per * copies == 0));
            for (size_t i = 0; i < skip; i++) o[i] = f[i];
            for (uint32_t j = 0; j < copies; j++) for (size_t i = 0; i < per; i++) o[skip + j * per + i] = f[skip + i];
            f.swap(o);
        };
        rep(a_aux, 0, cs->num_aux); rep(b_aux, 0, cs->num_aux); rep(b_in, 1, cs->num_input - 1);
    }
    r->n_table = table.size();
    if (hipMalloc((void **)&r->table, table.size() * sizeof(Fr)) != hipSuccess || hipMalloc((void **)&r->d_a_aux, a_aux.size()) != hipSuccess ||
        hipMalloc((void **)&r->d_b_in, b_in.size()) != hipSuccess || hipMalloc((void **)&r->d_b_aux, b_aux.size()) != hipSuccess) {
        ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM);
    }
    if (hipMemcpy(r->table, table.data(), table.size() * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->d_a_aux, a_aux.data(), a_aux.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->d_b_in, b_in.data(), b_in.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->d_b_aux, b_aux.data(), b_aux.size(), hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "r1cs: upload failed"; return fail(FK_ERR_HIP); }
    for (uint32_t j = 0; j < r->num_aux; j++) { r->n_a_aux += a_aux[j]; r->n_b_aux += b_aux[j]; }
    for (uint32_t i = 0; i < r->num_input; i++) r->n_b_in += b_in[i];
    {   // query index lists (bellman's order: inputs first, then the aux variables the density map selects)
        std::vector<uint32_t> ia, ib;
        ia.reserve(r->num_input + r->n_a_aux); ib.reserve(r->n_b_in + r->n_b_aux);
        for (uint32_t i = 0; i < r->num_input; i++) { ia.push_back(i); if (b_in[i]) ib.push_back(i); }
        for (uint32_t j = 0; j < r->num_aux; j++) { if (a_aux[j]) ia.push_back(r->num_input + j); if (b_aux[j]) ib.push_back(r->num_input + j); }
        if (hipMalloc((void **)&r->d_idx_a, ia.size() * 4 + 4) != hipSuccess || hipMalloc((void **)&r->d_idx_b, ib.size() * 4 + 4) != hipSuccess) {
            ctx->err = "r1cs: device allocation failed"; return fail(FK_ERR_OOM);
        }
        if ((ia.size() && hipMemcpy(r->d_idx_a, ia.data(), ia.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ||
            (ib.size() && hipMemcpy(r->d_idx_b, ib.data(), ib.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) { ctx->err = "r1cs: upload failed"; return fail(FK_ERR_HIP); }
        r->qidx.d_a_aux = r->d_a_aux; r->qidx.d_b_in = r->d_b_in; r->qidx.d_b_aux = r->d_b_aux;
        r->qidx.a = r->d_idx_a; r->qidx.b = r->d_idx_b; r->qidx.n_a = ia.size(); r->qidx.n_b = ib.size();
    }
    *out = r;
    return FK_OK;
}

int fk_r1cs_load(fk_ctx *ctx, const fk_r1cs *cs, fk_r1cs_dev **out) { return fk_guard(ctx, [&]() -> int { return r1cs_load_impl(ctx, cs, 1, out); }); }
}  // extern "C"
namespace fk {
int r1cs_load_coded(fk_ctx *ctx, uint32_t num_input, uint32_t num_aux, uint64_t num_gates, const uint64_t *const ptr[3], const uint32_t *const col[3],
                    const uint32_t *const cidx[3], const Fr *table, uint64_t n_table, fk_r1cs_dev **out, const uint8_t *const density[3]) {
    fk_r1cs cs{};
    cs.num_input = num_input; cs.num_aux = num_aux; cs.num_gates = num_gates;
    cs.a_ptr = ptr[0]; cs.a_col = col[0]; cs.b_ptr = ptr[1]; cs.b_col = col[1]; cs.c_ptr = ptr[2]; cs.c_col = col[2];
    return r1cs_load_impl(ctx, &cs, 1, out, cidx, table, n_table, density);
}
}  // namespace fk
extern "C" {
int fk_r1cs_load_tiled(fk_ctx *ctx, const fk_r1cs *instance, uint32_t copies, fk_r1cs_dev **out) { return fk_guard(ctx, [&]() -> int { return r1cs_load_impl(ctx, instance, copies, out); }); }

// the same system with its coefficients already dictionary-coded by the caller: *_val of `cs` are ignored, cidx[k][i] indexes
// `table` (n_table Montgomery values, table[0] = ONE).  8 bytes per term on the host side as well -- how a system of 10^9
// explicit terms is handed over (32-byte values per term would be 30 GB of host memory).
int fk_r1cs_load_coded(fk_ctx *ctx, const fk_r1cs *cs, const uint32_t *a_cidx, const uint32_t *b_cidx, const uint32_t *c_cidx, const uint64_t *table,
                       uint64_t n_table, fk_r1cs_dev **out) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!cs || !table || !n_table || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: null argument");
    const uint32_t *cidx[3] = {a_cidx, b_cidx, c_cidx};
    const uint64_t *ptrs[3] = {cs->a_ptr, cs->b_ptr, cs->c_ptr};
    for (int k = 0; k < 3; k++) if (ptrs[k] && ptrs[k][cs->num_gates] && !cidx[k]) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: null coefficient index array");
    return r1cs_load_impl(ctx, cs, 1, out, cidx, (const Fr *)table, n_table);
}); }

int fk_r1cs_density_ptrs(const fk_r1cs_dev *r, const void *out[3]) {
    if (!r || !out) return FK_ERR_BAD_ARG;
    out[0] = r->d_a_aux; out[1] = r->d_b_in; out[2] = r->d_b_aux;
    return FK_OK;
}

int fk_r1cs_info(const fk_r1cs_dev *r, uint64_t out[8]) {
    if (!r || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[8] = {r->num_gates + r->num_input, r->nnz[0], r->nnz[1], r->nnz[2], r->n_table,
                           (uint64_t)r->num_input + r->n_a_aux, r->n_b_in + r->n_b_aux, (uint64_t)r->num_input + r->num_aux};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

int fk_r1cs_windows(const fk_r1cs_dev *r, uint32_t *n_windows, uint64_t rows[17], uint64_t need[16]) {
    if (!r || !n_windows) return FK_ERR_BAD_ARG;
    *n_windows = r->win_k;
    for (uint32_t j = 0; j <= r->win_k && rows; j++) rows[j] = r->win_row[j];
    for (uint32_t j = 0; j < r->win_k && need; j++) need[j] = r->win_need[j];
    return FK_OK;
}

}  // extern "C"
namespace fk {
// The per-log_w residue-grouped class lists of a constraint system with binned matrices (see SliceLists), built on first use.
static int slice_lists(fk_ctx *ctx, const fk_r1cs_dev *r, uint32_t log_w, const SliceLists **out) {
    std::lock_guard<std::mutex> lock(r->slice_mu);
    SliceLists &sl = r->slices[log_w];
    if (!sl.built) {
        const uint32_t W = 1u << log_w;
        for (int k = 0; k < 3; k++) {
            if (!((r->bins.mask >> k) & 1)) continue;
            const std::vector<uint32_t> &src = r->h_rowlist[k];
            std::vector<uint32_t> dst(src.size());
            for (uint32_t s = 0; s < r->bins.nseg; s++) {
                if (r->bins.mtx[s] != (uint32_t)k) continue;
                const uint32_t off = r->bins.list_off[s], nr = r->bins.n_rows[s];
                uint32_t c[8] = {0};
                for (uint32_t i = 0; i < nr; i++) c[src[off + i] & (W - 1)]++;
                uint32_t pos[8], acc = 0;
                for (uint32_t q = 0; q < W; q++) { sl.cnt[s][q] = c[q]; sl.offs[s][q] = acc; pos[q] = acc; acc += c[q]; }
                for (uint32_t i = 0; i < nr; i++) { const uint32_t row = src[off + i]; dst[off + pos[row & (W - 1)]++] = row; }   // stable: longest first inside a group
            }
            FK_HIP(ctx, hipMalloc((void **)&sl.d_list[k], dst.size() * 4 + 4));
            if (!dst.empty()) FK_HIP(ctx, hipMemcpy(sl.d_list[k], dst.data(), dst.size() * 4, hipMemcpyHostToDevice));
        }
        sl.built = true;
    }
    *out = &sl;
    return FK_OK;
}

// a, b, c <- the rows t = rank (mod 2^log_w) of A z, B z, C z, densely: out[j] = row rank + j * 2^log_w (sliced), or all rows
// (not sliced: rank = log_w = 0).  n_out: elements the caller's arrays hold; those behind the last row are zeroed when sliced.
int r1cs_eval_impl(fk_ctx *ctx, const fk_r1cs_dev *r, const void *d_z, void *d_a, void *d_b, void *d_c, bool sliced, uint32_t rank, uint32_t log_w, uint64_t n_out, int window = -1) {
    FK_HIP(ctx, hipSetDevice(ctx->device));
    SpmvArgs a;
    for (int k = 0; k < 3; k++) { a.ptr[k] = r->ptr[k]; a.col[k] = r->col[k]; a.cidx[k] = r->cidx[k]; }
    a.out[0] = (Fr *)d_a; a.out[1] = (Fr *)d_b; a.out[2] = (Fr *)d_c;
    const uint64_t rows = r->num_gates + r->num_input;
    const uint64_t W = (uint64_t)1 << log_w;
    const uint64_t my_rows = sliced ? (rows > rank ? (rows - rank + W - 1) / W : 0) : rows;      // row groups this launch evaluates
    if (sliced) {
        if (my_rows > n_out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs eval: the slice holds %llu rows, the arrays %llu", (unsigned long long)my_rows, (unsigned long long)n_out);
        if (n_out > my_rows) for (int k = 0; k < 3; k++) FK_HIP(ctx, hipMemsetAsync(a.out[k] + my_rows, 0, (n_out - my_rows) * sizeof(Fr), ctx->stream));
    }
    // lanes per row: half of the matrix's mean row length, rounded down to a power of two (1 .. 64); measured on the eddsa batch: 16 / 8 / 4 / 2 / 1 terms per lane -> 1.51 / 1.27 / 1.16 / 1.02 / 0.97 ms
    uint32_t lg[3], lgmax = 0;
    static const int t_div = std::max(1, tune("FK_SPMV_TERMS_PER_LANE", 2));
    for (int k = 0; k < 3; k++) {
        const uint64_t mean = r->num_gates ? r->nnz[k] / r->num_gates : 0;
        lg[k] = 0;
        while (!((r->bins.mask >> k) & 1) && lg[k] < 6 && ((uint64_t)t_div << lg[k]) <= mean) lg[k]++;
        if (lg[k] > lgmax) lgmax = lg[k];
    }
    const uint64_t lanes = my_rows << lgmax;
    const TileDims td{r->base_gates, r->base_input, r->base_aux};
    const uint32_t binned = r->bins.mask;
    const dim3 grid((unsigned)((lanes + 255) / 256), 3), block(256);
    const bool tiled = r->copies > 1;
    // window >= 0 (explicit system, every matrix binned): only the class-list entries of row window `window`; the rows behind the gates
    // (this kernel's only work then) go with window 0
    if (lanes && window <= 0) {
#define FK_SPMV_LAUNCH(T_, S_) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_kernel<T_, S_>), grid, block, 0, ctx->stream, a, r->table, (const Fr *)d_z, r->num_gates, \
                                                  r->num_input, lg[0], lg[1], lg[2], td, binned, log_w, rank)
        if (tiled) { if (sliced) FK_SPMV_LAUNCH(true, true); else FK_SPMV_LAUNCH(true, false); }
        else { if (sliced) FK_SPMV_LAUNCH(false, true); else FK_SPMV_LAUNCH(false, false); }
#undef FK_SPMV_LAUNCH
    }
    if (binned) {
        BinArgs b = r->bins;
        SliceArgs sa;
        if (sliced) {
            // this rank's plan: copy c needs the rows = rank - c * base_gates (mod W); the residues repeat every P copies
            const SliceLists *sl = nullptr;
            FK_TRY(slice_lists(ctx, r, log_w, &sl));
            sa.log_w = log_w; sa.rank = rank;
            const uint32_t g = (uint32_t)(r->base_gates & (W - 1));
            uint32_t P = 1;
            while ((uint32_t)((uint64_t)P * g) & (W - 1)) P++;
            sa.P = P;
            for (uint32_t p_ = 0; p_ < P; p_++) sa.rho[p_] = (uint32_t)((rank + W * P - ((uint64_t)p_ * g & (W - 1))) & (W - 1));
            b.first_block[0] = 0;
            for (uint32_t s = 0; s < b.nseg; s++) {
                memcpy(sa.cnt[s], sl->cnt[s], sizeof sa.cnt[s]); memcpy(sa.offs[s], sl->offs[s], sizeof sa.offs[s]);
                uint64_t T = 0, part = 0;
                const uint32_t tail = r->copies % P;
                for (uint32_t p_ = 0; p_ < P; p_++) { T += sa.cnt[s][sa.rho[p_]]; if (p_ < tail) part += sa.cnt[s][sa.rho[p_]]; }
                sa.T[s] = (uint32_t)T;
                const uint64_t groups = (uint64_t)(r->copies / P) * T + part, per = 256u >> b.lg[s], blocks = (groups + per - 1) / per;
                if (b.first_block[s] + blocks > 0x7fffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs: system too large for the binned product");
                b.first_block[s + 1] = b.first_block[s] + (uint32_t)blocks;
            }
            for (int k = 0; k < 3; k++) b.rowlist[k] = sl->d_list[k];
        }
        if (tiled && !sliced && r->n_wavelist) {
            // rows of wave_lo .. wave_hi - 1 terms go to spmv_tiled_wave_kernel: they are rowlist[wave_from, wave_to) of their matrix
            // (the lists are sorted by length, longest first), i.e. with the default [4, 64): the whole 8-, 4- and 2-lane classes and the
            // 4 .. 7-term head of the one-lane class; with wave_hi > 64 also the tail of the 16-lane class
            b.first_block[0] = 0;
            for (uint32_t s = 0; s < b.nseg; s++) {
                const uint32_t k = b.mtx[s], s0 = b.list_off[s], s1 = s0 + b.n_rows[s], w0 = r->wave_from[k], w1 = r->wave_to[k];
                uint32_t lo = s0, hi = s1;
                if (w0 <= s0) lo = std::min(std::max(s0, w1), s1); else hi = std::min(s1, w0);      // (load keeps the wave range from lying strictly inside a class)
                b.list_off[s] = lo; b.n_rows[s] = hi - lo;
                const uint64_t groups = (uint64_t)b.n_rows[s] * r->copies, per = 256u >> b.lg[s];
                b.first_block[s + 1] = b.first_block[s] + (uint32_t)((groups + per - 1) / per);
            }
            const uint64_t n_waves = (uint64_t)((r->copies + 63) / 64) * r->n_wavelist;
            hipLaunchKernelGGL(spmv_tiled_wave_kernel, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, ctx->stream, a, r->table, (const Fr *)d_z, r->num_input, td,
                               r->copies, r->d_wavelist, r->n_wavelist, (uint32_t)n_waves);
        }
        if (window >= 0) {
            b.first_block[0] = 0;
            for (uint32_t s = 0; s < b.nseg; s++) {
                const uint32_t c0 = r->win_cnt[window][s], c1 = r->win_cnt[window + 1][s];
                b.list_off[s] += c0; b.n_rows[s] = c1 - c0;
                const uint64_t per = 256u >> b.lg[s];
                b.first_block[s + 1] = b.first_block[s] + (uint32_t)((b.n_rows[s] + per - 1) / per);
            }
        }
        const unsigned blocks = b.first_block[b.nseg];
        if (blocks) {
#define FK_SPMVB_LAUNCH(T_, S_) hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_binned_kernel<T_, S_>), dim3(blocks), dim3(256), 0, ctx->stream, a, b, r->table, (const Fr *)d_z, \
                                                   r->num_input, td, r->copies, sa)
#ifdef FK_EXPERIMENTS
            if (!tiled && !sliced && tune("FK_SPMV_FAKE4", 0) && r->n_table > 4096)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_binned_kernel<false, false, 1>), dim3(blocks), dim3(256), 0, ctx->stream, a, b, r->table, (const Fr *)d_z,
                                   r->num_input, td, r->copies, sa);
            else if (!tiled && !sliced && window < 0 && tune("FK_SPMV_SEQ", 0) && b.pptr[0] && b.pptr[1] && b.pptr[2])
                hipLaunchKernelGGL(HIP_KERNEL_NAME(spmv_binned_kernel<false, false, 2>), dim3(blocks), dim3(256), 0, ctx->stream, a, b, r->table, (const Fr *)d_z,
                                   r->num_input, td, r->copies, sa);
            else
#endif
            if (tiled) { if (sliced) FK_SPMVB_LAUNCH(true, true); else FK_SPMVB_LAUNCH(true, false); }
            else { if (sliced) FK_SPMVB_LAUNCH(false, true); else FK_SPMVB_LAUNCH(false, false); }
#undef FK_SPMVB_LAUNCH
        }
    }
    FK_HIP(ctx, hipGetLastError());
    FK_DBG(ctx, "spmv");
    return FK_OK;
}
}  // namespace fk
extern "C" {

// a, b, c: device arrays with room for next_pow2(rows) elements each (fk_prove_dev's contract); rows written.
int fk_r1cs_eval_dev(fk_ctx *ctx, const fk_r1cs_dev *r, const void *d_z, void *d_a, void *d_b, void *d_c) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!r || !d_z || !d_a || !d_b || !d_c) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs eval: null argument");
    return r1cs_eval_impl(ctx, r, d_z, d_a, d_b, d_c, false, 0, 0, 0);
}); }

// Multi-GPU form: only the cyclic slice rank `rank` of 2^log_w ranks needs -- local[j] = row (rank + j * 2^log_w) of A z, B z,
// C z, zero behind the last row: exactly what fk_dq_gather_dev would cut out of the full vectors, at 1 / 2^log_w of the work and
// without the three m-element vectors.  Arrays of 2^(log_m - log_w) elements.
int fk_r1cs_eval_slice_dev(fk_ctx *ctx, const fk_r1cs_dev *r, const void *d_z, uint32_t log_m, uint32_t rank, uint32_t log_w, void *d_a, void *d_b, void *d_c) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!r || !d_z || !d_a || !d_b || !d_c) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs eval: null argument");
    if (log_w > 3 || (rank >> log_w) || log_m < log_w || log_m >= FK_FR_S) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs eval: bad slice (rank %u of 2^%u, domain 2^%u)", rank, log_w, log_m);
    if (r->num_gates + r->num_input > ((uint64_t)1 << log_m)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "r1cs eval: the system has more rows than the domain 2^%u", log_m);
    return r1cs_eval_impl(ctx, r, d_z, d_a, d_b, d_c, true, rank, log_w, (uint64_t)1 << (log_m - log_w));
}); }

// witness in -> proof out: SpMV, quotient, MSMs, assembly.  z: device pointer (num_input + num_aux elements).
int fk_prove_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const void *d_z, const uint64_t rr[4], const uint64_t ss[4],
                      uint8_t out_proof[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_prove_r1cs_dev");
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !r || !d_z) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (r->num_input != key->num_input || r->num_aux != key->num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: constraint system and key disagree on the variable counts");
    const uint64_t rows = r->num_gates + r->num_input;
    if (rows > key->m || (key->m > 1 && rows <= key->m / 2)) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: %llu rows do not match key domain %llu",
                                                                     (unsigned long long)rows, (unsigned long long)key->m);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t mb = key->m * sizeof(Fr);
    if (ctx->early.done && ctx->early.key == key && ctx->early.r1cs == r && ctx->early.d_z == d_z) {
        // the front of this proof -- ev_z, the evaluation of a, b, c into the stage buffers, the witness multiplications' sorts --
        // was queued while the previous proof's tails ran (early_front below): go on with the quotient
        ctx->qidx = &r->qidx;
        ctx->ev_z_recorded = true;
        const int rc = fk_prove_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, rows, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux, rr, ss, out_proof, tm);
        ctx->qidx = nullptr;
        ctx->ev_z_recorded = false;
        return rc;
    }
    if (ctx->early.done) { msm_abandon(ctx); FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: an early front of another proof is outstanding"); }
    FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
    // z is complete at this point of the main stream: the witness multiplications wait for THIS, not for the evaluation of a, b, c
    // (11.6 ms at 2^25 during which nothing else ran; FK_PROVE_Z_EARLY=0 restores that)
    static const int t_zearly = tune("FK_PROVE_Z_EARLY", 1);
    if (t_zearly) {
        if (!ctx->ev_z) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
        FK_HIP(ctx, hipEventRecord(ctx->ev_z, ctx->stream));
        ctx->ev_z_recorded = true;
    }
    ctx->qidx = &r->qidx;      // the queries' index lists are known: no per-proof density compaction
    const int rce = fk_r1cs_eval_dev(ctx, r, d_z, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p);
    if (rce != FK_OK) { ctx->qidx = nullptr; ctx->ev_z_recorded = false; return rce; }
    const int rc = fk_prove_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, rows, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux, rr, ss, out_proof, tm);
    ctx->qidx = nullptr;
    ctx->ev_z_recorded = false;
    return rc;
}); }

// multi-GPU: all five multiplications of this key's slices for a resident constraint system (see fk_prove_msms_hz_dev)
int fk_prove_msms_hz_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const void *d_h_slice, const void *d_z,
                              uint8_t out[FK_MSM_RESULT_BYTES]) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !r || !d_z) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (r->num_input != key->num_input || r->num_aux != key->num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: constraint system and key disagree on the variable counts");
    ctx->qidx = &r->qidx;
    const int rc = fk_prove_msms_hz_dev(ctx, key, d_h_slice, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux, out, nullptr);
    ctx->qidx = nullptr;
    return rc;
}); }

// ... and of fk_prove_msms_z_begin_dev: the witness multiplications are queued (index-list gathers), the caller runs the
// (distributed) quotient and finishes with fk_prove_msms_finish_dev
int fk_prove_msms_z_begin_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const void *d_z) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !r || !d_z) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    if (r->num_input != key->num_input || r->num_aux != key->num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: constraint system and key disagree on the variable counts");
    ctx->qidx = &r->qidx;
    const int rc = fk_prove_msms_z_begin_dev(ctx, key, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux);
    ctx->qidx = nullptr;
    return rc;
}); }

}  // extern "C"
namespace fk {
// ONE proof at a time from a witness in host memory, with the hand-over CHUNKED (round 5): the 1.07 GB upload of the benchmark's witness is
// 19 ms during which nothing else of this proof could run -- the proof's latency was upload + proof.  A circuit's rows read the variables
// allocated before them, so the rows below win_row[j + 1] need only the first win_need[j] elements of z (planned at load, r1cs.hpp): the
// witness goes up in win_k pieces on the copy stream and the evaluation of window j is queued behind piece j -- a, b, c are complete ~2 ms after
// the last byte has landed instead of ~20 ms, and the witness sorts then run without the evaluation beside them.  A system whose first rows
// read late variables degenerates to "upload everything, then evaluate" (win_need[0] = all of z).  Same kernels, same row order inside a
// window, same bytes.
static int prove_r1cs_chunked(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const uint64_t *z, const uint64_t rr[4], const uint64_t ss[4],
                              uint8_t out_proof[FK_PROOF_BYTES], fk_timings *tm) {
    const uint64_t rows = r->num_gates + r->num_input;
    const size_t mb = key->m * sizeof(Fr);
    FK_HIP(ctx, ctx->stage_a.reserve(mb)); FK_HIP(ctx, ctx->stage_b.reserve(mb)); FK_HIP(ctx, ctx->stage_c.reserve(mb));
    if (!ctx->copy_st) FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_st, hipStreamNonBlocking));
    if (!ctx->ev_upload_gate) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload_gate, hipEventDisableTiming));
    if (!ctx->ev_z) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
    // the pieces follow whatever the main stream still has queued (stage_z may be read by it)
    FK_HIP(ctx, hipEventRecord(ctx->ev_upload_gate, ctx->stream));
    FK_HIP(ctx, hipStreamWaitEvent(ctx->copy_st, ctx->ev_upload_gate, 0));
    Fr *d_z = ctx->stage_z.as<Fr>();
    uint64_t off = 0;
    ctx->qidx = &r->qidx;
    int rc = FK_OK;
    // piece j, then window j: from pageable memory hipMemcpyAsync returns only when the piece has been staged, so the windows must be
    // queued BETWEEN the pieces for the evaluation to run beside the rest of the upload
    for (uint32_t j = 0; j < r->win_k && rc == FK_OK; j++) {
        const uint64_t end = r->win_need[j];
        hipError_t e = hipSuccess;
        if (end > off) e = hipMemcpyAsync(d_z + off, z + 4 * off, (end - off) * sizeof(Fr), hipMemcpyHostToDevice, ctx->copy_st);
        if (e == hipSuccess && !ctx->ev_chunk[j]) e = hipEventCreateWithFlags(&ctx->ev_chunk[j], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_chunk[j], ctx->copy_st);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_chunk[j], 0);
        off = std::max(off, end);
        if (e == hipSuccess && j + 1 == r->win_k) {        // z is complete at this point of the main stream: the witness multiplications wait for THIS, not for the last window
            e = hipEventRecord(ctx->ev_z, ctx->stream);
            ctx->ev_z_recorded = e == hipSuccess;
        }
        if (e != hipSuccess) { rc = FK_ERR_HIP; ctx->err = std::string("prove: chunked hand-over: ") + hipGetErrorString(e); break; }
        rc = r1cs_eval_impl(ctx, r, d_z, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, false, 0, 0, 0, (int)j);
    }
    if (rc == FK_OK) rc = fk_prove_dev(ctx, key, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p, rows, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux, rr, ss, out_proof, tm);
    else { (void)hipStreamSynchronize(ctx->copy_st); (void)hipStreamSynchronize(ctx->stream); }        // the caller's buffer is no longer read when an error returns
    ctx->qidx = nullptr;
    ctx->ev_z_recorded = false;
    return rc;
}
}  // namespace fk
extern "C" {

int fk_prove_r1cs(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const uint64_t *z, const uint64_t rr[4], const uint64_t ss[4],
                  uint8_t out_proof[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_prove_r1cs");
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !r || !z) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t zb = ((size_t)r->num_input + r->num_aux) * sizeof(Fr);
    FK_HIP(ctx, ctx->stage_z.reserve(zb));
    static const bool t_chunked = !(getenv("FK_PROVE_CHUNKED_UPLOAD") && atoi(getenv("FK_PROVE_CHUNKED_UPLOAD")) == 0);      // documented switch (fawkes_hip.h)
    if (t_chunked && r->win_k >= 2 && !ctx->early.done && !ctx->wslot[0].pending && !ctx->wslot[1].pending) {
        if (r->num_input != key->num_input || r->num_aux != key->num_aux) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: constraint system and key disagree on the variable counts");
        const uint64_t rows = r->num_gates + r->num_input;
        if (rows > key->m || (key->m > 1 && rows <= key->m / 2)) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "prove: %llu rows do not match key domain %llu",
                                                                         (unsigned long long)rows, (unsigned long long)key->m);
        return prove_r1cs_chunked(ctx, key, r, z, rr, ss, out_proof, tm);
    }
    FK_HIP(ctx, hipMemcpyAsync(ctx->stage_z.p, z, zb, hipMemcpyHostToDevice, ctx->stream));
    return fk_prove_r1cs_dev(ctx, key, r, ctx->stage_z.p, rr, ss, out_proof, tm);
}); }

// Pipelined form of fk_prove_r1cs: _submit starts the upload of the witness into one of the two slots and returns at once;
// _wait computes that proof.  Calling submit(k+1) before wait(k) hides the upload of proof k+1 underneath proof k.
int fk_prove_r1cs_submit(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r, const uint64_t *z, const uint64_t rr[4], const uint64_t ss[4], int *ticket) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !r || !z || !rr || !ss || !ticket) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: null argument");
    const int slot = ctx->wslot_next;
    if (ctx->wslot[slot].pending) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: two proofs are already submitted (call fk_prove_r1cs_wait first)");
    fk_ctx::WitSlot &w = ctx->wslot[slot];
    const size_t zb = ((size_t)r->num_input + r->num_aux) * sizeof(Fr);
    // With another proof submitted ahead of this one, the upload is left to THAT proof's run: it queues the copy behind its
    // memory-bound front (sorts, evaluation of a, b, c), so that the transfer runs underneath transforms and accumulations.
    // Started here it ran beside the sorts: +14 ms per proof on the synthetic 2^25 shape (1 GiB witness), +0.8 ms on the
    // 1024-transaction system (profiles/r02_sorts_first_probe.log).  FK_UPLOAD_DEFER=0: start it here.  The host buffer must
    // stay valid until fk_prove_r1cs_wait(ticket) returns either way.
    static const int t_defer = tune("FK_UPLOAD_DEFER", 1);
    if (t_defer && ctx->wslot[slot ^ 1].pending && zb <= w.buf.cap && w.ready) { w.deferred = true; w.host_z = z; w.host_bytes = zb; }
    else { w.deferred = false; FK_TRY(fk_witness_upload_async(ctx, slot, z, zb)); }
    w.pending = true; w.key = key; w.r1cs = r;
    memcpy(w.r, rr, 32); memcpy(w.s, ss, 32);
    ctx->wslot_next = slot ^ 1;
    *ticket = slot;
    return FK_OK;
}); }
}  // extern "C"
namespace fk {
// Early front of the proof waiting in witness slot `slot` (pipelined proofs at sizes that run the sorts-first schedule): called by
// the prover when everything of the CURRENT proof is queued.  The current proof ends on latency-bound tails -- G2's bucket
// reduction behind H's accumulation: ~7 ms at 2^25 during which the GPU is mostly idle -- and the next proof begins with
// memory-bound work that needs nothing from it: the evaluation of a, b, c and the witness multiplications' sorts.  They are queued
// here behind the current proof's LAST ACCUMULATION (underneath an accumulation they would only take its time, csrc/prover.hip),
// so they fill that idle stretch and the host's turnaround between two proofs.
static int early_front(fk_ctx *ctx, int slot) {
    fk_ctx::WitSlot &w = ctx->wslot[slot];
    const fk_key *key = w.key; const fk_r1cs_dev *r = w.r1cs;
    if (w.deferred) { w.deferred = false; FK_TRY(fk_witness_upload_async(ctx, slot, w.host_z, w.host_bytes)); }
    void *d_z = nullptr;
    FK_TRY(fk_witness_ptr(ctx, slot, &d_z));               // the main stream waits for the slot's upload
    // FK_PROVE_EARLY_SORTS (experiment build): the witness sorts are queued BEFORE the wait for H's accumulation -- each lane's stream orders them
    // behind that lane's own work of the current proof -- and only the evaluation waits
    static const int t_early_sorts = tune("FK_PROVE_EARLY_SORTS", 0);
    if (!t_early_sorts && ctx->ev_acc_done_valid) FK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_acc_done, 0));      // ... and for the current proof's last accumulation (H's)
    const size_t mb = key->m * sizeof(Fr);
    if (mb > ctx->stage_a.cap || mb > ctx->stage_b.cap || mb > ctx->stage_c.cap) return FK_OK;       // never grow buffers the current proof may still read: no early front
    if (!ctx->ev_z) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
    FK_HIP(ctx, hipEventRecord(ctx->ev_z, ctx->stream));
    ctx->qidx = &r->qidx;
    int tails[4];
    const int wb = early_witness_begin(ctx, key, d_z, r->d_a_aux, r->d_b_in, r->d_b_aux, tails);
    if (wb < 0) { ctx->qidx = nullptr; return -wb; }
    if (wb == 0) { ctx->qidx = nullptr; return FK_OK; }     // (the schedule does not apply: nothing was queued but the waits)
    if (t_early_sorts && ctx->ev_acc_done_valid) FK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_acc_done, 0));
    const int rce = fk_r1cs_eval_dev(ctx, r, d_z, ctx->stage_a.p, ctx->stage_b.p, ctx->stage_c.p);
    ctx->qidx = nullptr;
    if (rce != FK_OK) return rce;
    ctx->early.done = true; ctx->early.key = key; ctx->early.r1cs = r; ctx->early.d_z = d_z;
    memcpy(ctx->early.tails, tails, sizeof tails);
    return FK_OK;
}
}  // namespace fk
extern "C" {
int fk_prove_r1cs_wait(fk_ctx *ctx, int ticket, uint8_t out_proof[FK_PROOF_BYTES], fk_timings *tm) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_prove_r1cs_wait");
    if (!ctx) return FK_ERR_BAD_ARG;
    if (ticket < 0 || ticket > 1 || !ctx->wslot[ticket].pending) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "prove: no submitted proof with ticket %d", ticket);
    fk_ctx::WitSlot &w = ctx->wslot[ticket];
    w.pending = false;
    if (w.deferred) { w.deferred = false; FK_TRY(fk_witness_upload_async(ctx, ticket, w.host_z, w.host_bytes)); }      // nobody ran in between
    void *d_z = nullptr;
    const bool early = ctx->early.done && ctx->early.key == w.key && ctx->early.r1cs == w.r1cs && ctx->early.d_z == ctx->wslot[ticket].buf.p;
    if (early) d_z = ctx->wslot[ticket].buf.p;            // its front is queued already (the main stream waited for the upload there)
    else FK_TRY(fk_witness_ptr(ctx, ticket, &d_z));
    // the other slot holds the NEXT proof of the same key and system: its front goes behind this proof's last accumulation
    const fk_ctx::WitSlot &o = ctx->wslot[ticket ^ 1];
    ctx->before_block = nullptr;
    if (o.pending && o.key == w.key && o.r1cs == w.r1cs && early_front_applies(w.key)) {
        const int other = ticket ^ 1;
        ctx->before_block = [ctx, other]() { return early_front(ctx, other); };
    }
    const int rc = fk_prove_r1cs_dev(ctx, w.key, w.r1cs, d_z, w.r, w.s, out_proof, tm);
    ctx->before_block = nullptr;
    return rc;
}); }

}  // extern "C"
