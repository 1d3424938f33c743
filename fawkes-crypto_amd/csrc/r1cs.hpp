// The device-resident constraint system (spmv.hip builds and evaluates it; multi.hip keeps one replica per GPU).
#pragma once
#include "common.hpp"
#include <mutex>

namespace fk {
// launch plan of spmv_binned_kernel: up to SPMV_CLASSES length classes for each of the 3 matrices
static constexpr int SPMV_CLASSES = 5, SPMV_SEGS = 3 * SPMV_CLASSES;
struct BinArgs {
    uint32_t nseg = 0, mask = 0;            // mask: bit k = matrix k is binned
    uint32_t first_block[SPMV_SEGS + 1] = {0}, lg[SPMV_SEGS] = {0}, mtx[SPMV_SEGS] = {0}, n_rows[SPMV_SEGS] = {0}, list_off[SPMV_SEGS] = {0};
    const uint32_t *rowlist[3] = {nullptr, nullptr, nullptr};
    const uint64_t *pptr[3] = {nullptr, nullptr, nullptr};      // experiment builds, FK_SPMV_SEQ: row pointers of a layout permuted into class-list order
};
// Cyclic row slices (multi-GPU: rank g of W = 2^log_w evaluates only the rows t = g (mod W), the slice the distributed
// quotient starts from).  The length-class lists are kept a second time per log_w with every class grouped by row mod W
// (inside a group still longest first), so that a rank's rows of a class are one contiguous run per copy; which residue a
// copy needs depends on the copy (row t = copy * base_gates + row), with period P = W / gcd(base_gates, W) copies.
struct SliceLists {
    uint32_t *d_list[3] = {nullptr, nullptr, nullptr};    // per matrix, same length and class boundaries as rowlist
    uint32_t cnt[SPMV_SEGS][8] = {{0}}, offs[SPMV_SEGS][8] = {{0}};       // per segment: rows of each residue / their first position in the class
    bool built = false;
};
struct SliceArgs {
    uint32_t log_w = 0, rank = 0, P = 1;
    uint32_t rho[8] = {0};                 // residue copy q*P + p needs
    uint32_t T[SPMV_SEGS] = {0};           // groups of one period of copies, per segment
    uint32_t cnt[SPMV_SEGS][8] = {{0}}, offs[SPMV_SEGS][8] = {{0}};
};
}  // namespace fk

struct fk_r1cs_dev {
    uint32_t num_input = 0, num_aux = 0;
    uint64_t num_gates = 0;
    uint64_t *ptr[3] = {nullptr, nullptr, nullptr};
    uint32_t *col[3] = {nullptr, nullptr, nullptr};
    uint32_t *cidx[3] = {nullptr, nullptr, nullptr};
    fk::Fr *table = nullptr;
    uint64_t n_table = 0, nnz[3] = {0, 0, 0};
    uint8_t *d_a_aux = nullptr, *d_b_in = nullptr, *d_b_aux = nullptr;
    uint64_t n_a_aux = 0, n_b_in = 0, n_b_aux = 0;   // popcounts
    uint32_t *d_idx_a = nullptr, *d_idx_b = nullptr; // variables of the A / B query in query order
    fk::QueryIdx qidx;
    // tiled system (fk_r1cs_load_tiled): the CSR above is ONE instance (base_gates rows, base_input / base_aux variables)
    // and stands for `copies` of it; num_input / num_aux / num_gates / nnz are the totals
    uint32_t copies = 1, base_input = 0, base_aux = 0, base_gates = 0;
    // matrices with long rows: the (instance's) rows in classes by length, sorted by length inside a class (see spmv_binned_kernel)
    uint32_t *rowlist[3] = {nullptr, nullptr, nullptr};
    uint64_t *pptr[3] = {nullptr, nullptr, nullptr};             // (experiment builds only; freed with the system)
    fk::BinArgs bins;
    // batch circuits of >= 64 copies: the rows of middling length that spmv_tiled_wave_kernel takes (matrix << 30 | row, circuit
    // order) and where they sit in rowlist (sorted by length: [wave_from, wave_to) of each matrix)
    uint32_t *d_wavelist = nullptr, n_wavelist = 0, wave_from[3] = {0, 0, 0}, wave_to[3] = {0, 0, 0};
    // Row windows (explicit systems whose class lists are block-sorted, all three matrices binned): the gate rows cut into win_k runs of
    // consecutive rows.  win_cnt[j][s]: entries of launch segment s whose row lies below win_row[j] (a PREFIX of the segment's list: the
    // lists are ordered by blocks of consecutive rows); win_need[j]: how many leading elements of z the rows below win_row[j + 1] read.
    // fk_prove_r1cs uploads z in those pieces and evaluates window j as soon as piece j has landed (spmv.hip: prove_r1cs_chunked).
    static constexpr uint32_t WIN_MAX = 16;
    uint32_t win_k = 0;
    uint64_t win_row[WIN_MAX + 1] = {0}, win_need[WIN_MAX] = {0};
    uint32_t win_cnt[WIN_MAX + 1][fk::SPMV_SEGS] = {{0}};
    // host copies of the class lists and the per-log_w residue-grouped variants, built on first use (fk_r1cs_eval_slice_dev)
    std::vector<uint32_t> h_rowlist[3];
    mutable fk::SliceLists slices[4];
    mutable std::mutex slice_mu;
};
