// BN254 G1 / G2 Pippenger multi-scalar multiplication on gfx950.
//
// Replaces bellman_ce::multiexp::multiexp as called five times by bellman's prover (H, L, A, B1 in G1 and
// B2 in G2; reached from
// /root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80).  Any correct MSM yields the
// same group element, so the result is bit-identical to bellman's after `into_affine`.
//
// MI355X pipeline (no MFMA -- 256-bit modular integer work).  A multiplication is queued on one of MSM_LANES lanes (stream +
// private scratch); consecutive multiplications use different lanes, so the latency-bound steps 4-5 of one run underneath
// steps 1-3 of the next (msm_*_begin / msm_*_end).  Its front (steps 1-2b), its accumulation (3) and its tail (4-5) can be
// queued separately (fk_ctx::defer_back, msm_run_deferred): the prover's sorts-first schedule puts the quotient between the
// witness multiplications' fronts and their accumulations (prover.hip).
//   1. msm_digits      scalars leave Montgomery form (bellman's `into_repr`) and are cut into W signed digits; the 255
//                      bits are split evenly into windows of cb or cb + 1 bits (no short top window).
//   2. bucket sort     two-pass radix sort on the bucket index (up to 1024 high bins, then the low 10-11 bits inside each
//                      segment): LDS histograms and cursors, every tile counted, laid out in bin order in LDS and written
//                      as contiguous runs; 8192-entry tiles, two workgroups per compute unit in both passes.  No global atomics, so skewed witness scalars (many 0/1) cost nothing extra.
//                      The second pass is launched over the host's upper bound on the tile count; the cap of a bucket-lane,
//                      the list of oversized buckets and their segment tasks are worked out by small kernels into a
//                      device-side record (MsmDyn) the later kernels read -- the host never waits inside a multiplication.
//   2b. size order     buckets are counting-sorted by length so the 64 lanes of a wave run equally long.
//   3. msm_accumulate  one lane per bucket walks its run with XYZZ mixed additions (6 products, a dual squaring and one
//                      fused difference of two products, lazily reduced: field.hpp), gathering 64-byte affine bases; buckets above `cap` = mean + 6 sigma + 8 entries hand the excess to
//   4. msm_overflow    one wave per segment (sized so that all oversized entries give ~2 waves per SIMD), reduced with
//                      wavefront shuffles; then 256 lanes per oversized bucket fold the partials.
//   5. msm_bucket_reduce  sum_b b*S_b per window: each lane runs the running-sum trick over L buckets,
//                      adds its offset multiple by double-and-add, then wave64 shuffle + LDS reduction; a last small
//                      kernel folds the workgroups' partial sums, the window sums are copied to pinned host memory.
//   6. host            W window sums are Horner-combined (cw doublings each) -- microseconds.
#include "common.hpp"
#include <algorithm>
#include <string.h>
#include <type_traits>

namespace fk {

static constexpr uint32_t SEG_MAX = 4096, SEG_MIN = 256;   // overflow segment (entries) per wave: 4..64 per lane, sized per call
static constexpr uint32_t SORT_THREADS = 256;     // first-pass histogram: small workgroups are placed more easily underneath an accumulation (profiles/r01_corun_experiment.log)

struct MsmPlan {
    size_t n;
    uint32_t c, W, B;        // widest window's bits, windows, buckets per window (2^(c-1))
    uint32_t cb, wide;       // window widths: the first `wide` windows have cb + 1 bits, the others cb (sum = 255)
    uint32_t nchunks;
    size_t chunk;
    uint32_t cap, cap_top;   // max entries a bucket-lane handles itself (all windows but the last / the last, shorter one)
    uint32_t L, T, nblk;     // bucket-reduce: buckets per lane, lanes per window, blocks per window
    uint32_t LB, nhi, nlo;   // low bits of the bucket index, number of high / low bins
};
#ifndef FK_S2_TILE
#define FK_S2_TILE 8192
#endif
// entries per tile of the SECOND sort pass.  8192: two to three workgroups of its scatter kernel per compute unit (56 KB of LDS
// each) overlap each other's load and barrier waits -- the kernel's waves are parked 70 % of the time; 175.9 -> 174.6 ms per
// proof against 16384, 4096 is slower again (profiles/r02_sorts_first_probe.log).  The first pass keeps 16384 (S1_TILE).
static constexpr uint32_t S2_TILE = FK_S2_TILE;
static constexpr uint32_t S2_EPT = S2_TILE / 1024;      // entries per lane (1024-lane workgroups)
// Sub-tile of the FIRST pass (internal to its scatter kernel) and the number of high bins its LDS tables hold.  8192 entries and
// 1024 bins (the second pass then takes 11 low bits of a 21-bit bucket index): 76 KB of LDS, TWO workgroups per compute unit --
// the kernel's waves are parked 79 % of the time (loads, barriers), a second workgroup fills that: 173.9 -> 171.3 ms per proof
// against 16384 entries / 2048 bins (one workgroup per compute unit); 4096 entries (three per unit) 172.5
// (profiles/r02_sorts_first_probe.log).
#ifndef FK_S1_TILE
#define FK_S1_TILE 8192
#endif
static constexpr uint32_t S1_TILE = FK_S1_TILE;
#ifndef FK_S2_MAX_HI
#define FK_S2_MAX_HI 1024
#endif
static constexpr uint32_t S2_MAX_HI = FK_S2_MAX_HI;     // high bins of the first pass (make_plan gives the second pass the remaining low bits)

// Windows are sized from n.  Sizing them from the number of non-trivial scalars (0 and 1 never reach the ordinary
// buckets) was measured slower on the witness MSMs: fewer windows win even at bucket loads of ~6 once lanes are
// size-ordered (G2 accumulate 22.6 ms at c = 20 vs 26.2 ms at c = 16 for 16.7M scalars of which 3.3M are dense).
// merged: the bases carry precomputed levels (KeyPre), all windows share ONE bucket set -- the reduction costs a W-th, so the
// windows can be wider (FK_MSM_PRE_DC: +3 bits, up to the sort's limit of 22)
static MsmPlan make_plan(size_t n, unsigned forced_c, bool merged = false) {
    MsmPlan p{};
    p.n = n;
    const size_t nd = n ? n : 1;
    uint32_t lg = 0; while (((size_t)1 << (lg + 1)) <= nd + nd / 2) lg++;     // log2 rounded (2^25 - 1 counts as 2^25)
    // large MSMs: bucket loads of ~64 are enough now that lanes are size-ordered, so c grows with n (fewer digits
    // per scalar: 13 at c = 20 instead of 16)
    // tuning knobs: compile-time constants in a release build, read from the environment in an -DFK_EXPERIMENTS build (common.hpp: tune)
    static const int t_small = tune("FK_MSM_C_SMALL", 17), t_delta = tune("FK_MSM_C_DELTA", 5), t_sig = tune("FK_MSM_CAP_SIGMA", 6), t_pre_dc = tune("FK_MSM_PRE_DC", 3);
    uint32_t c = forced_c ? forced_c : (lg >= 22 ? lg - t_delta : (lg >= 18 ? (uint32_t)t_small : (lg >= 6 ? lg - 2 : 4)));
    if (merged && !forced_c) c = (uint32_t)std::max(2, (int)c + t_pre_dc);
    if (c < 2) c = 2;
    static const uint32_t t_cmax = (uint32_t)std::min(tune("FK_MSM_C_MAX", 22), S2_MAX_HI >= 2048 ? 24 : 22);     // the sort's limit: 2^(c - 1) buckets = S2_MAX_HI high bins x <= 4096 low bins
    if (c > t_cmax) c = t_cmax;
    p.c = c;
    p.W = (255 + c - 1) / c;
    p.cb = 255 / p.W;
    p.wide = 255 - p.cb * p.W;
    if (p.wide == 0) { p.cb -= 1; p.wide = p.W; }     // all windows equally wide
    c = p.cb + 1;                                      // the widest window actually used (e.g. a request of 21 gives 13 windows of <= 20 bits)
    p.c = c;
    p.B = 1u << (c - 1);
    size_t chunk = (n + 255) / 256;          // first-pass chunks per window
    if (chunk < 16384) chunk = 16384;
    p.chunk = chunk;
    p.nchunks = (uint32_t)((n + chunk - 1) / chunk);
    // cap = mean load of the NARROW windows (2^(cb-1) buckets in use) + 6 sigma (Poisson) + 8; longer buckets go through the
    // oversized-bucket path (64 lanes per segment).  One lane walks its bucket serially (~11 us per G1 addition, ~31 us
    // per G2 addition at the occupancy these kernels run at), so the cap is a floor on the accumulate kernel's duration:
    // with the former 2*mean + 64 the G2 kernel at 2^22 took cap x 31 us = 6 ms for 3 ms of work.  The windows split the
    // 255 bits evenly, so there is no short top window any more whose few buckets would all end up on that path.
    size_t mean = nd >> (p.wide < p.W ? p.cb - 1 : p.cb);
    p.cap = (uint32_t)std::min<size_t>(2 * mean + 64, 1u << 30);
    if (t_sig > 0) { uint32_t sq = 1; while ((size_t)sq * sq < mean) sq++; p.cap = (uint32_t)std::min<size_t>(mean + (size_t)t_sig * sq + 8, 1u << 30); }
    p.cap_top = p.cap;
    // bucket reduction: buckets per lane (<= 64).  L = 8 everywhere below 2^18 buckets (shorter serial chains) was measured
    // neutral-to-worse: the reduction already runs underneath the next multiplication.  FK_MSM_RED_L overrides (tuning).
    static const int t_redl = tune("FK_MSM_RED_L", 0);
    p.L = p.B >= (1u << 18) ? 64 : (p.B >= (1u << 17) ? p.B / 2048 : (p.B >= 8192 ? p.B / 4096 : 1));      // 2^16 buckets: 16 per lane (32: 2^20 13.0 -> 12.0 ms per proof, 8: 15.4); 2^17: 64 (32: 2^23 44.1 -> 45.1)
    if (t_redl > 0 && (uint32_t)t_redl <= p.B) p.L = (uint32_t)t_redl;
    // merged form: ONE bucket set, so the reduction is B / L lanes in all -- 2^19 buckets / 64 = 32 workgroups, a serial chain of 64 full
    // additions each, on the tail of every multiplication.  A shorter chain costs more additions in all (every lane also multiplies
    // its partial sum by its offset) and less time: FK_MSM_RED_L_MERGED (experiment builds), default by the round-4 sweeps below.
    // Round-4 sweeps (profiles/r04_small_levels_sweep.log, r04_shard_levels_sweep.log, r04_red_l_sweep.log): 2^19 buckets -- L = 8 (proof of
    // a 2^22 system 21.3 -> 20.6 ms, the share of rank 0 of 8 at 2^25 41.6 -> 39.7 ms), 2^20 buckets -- L = 16 (512 transactions 95.5 -> 89.6 ms),
    // 2^21 buckets (the benchmark size) -- L = 64 stays (223.0 / 224.1 ms per proof against 225.6 / 226.1 at 32, 227.6 at 16, 230 at 8:
    // there the reductions run underneath the next accumulation and only their work counts).
    static const int t_redl_m = tune("FK_MSM_RED_L_MERGED", 0);
    if (merged) {
        if (p.B <= (1u << 19)) p.L = std::min<uint32_t>(p.L, 8);
        else if (p.B <= (1u << 20)) p.L = std::min<uint32_t>(p.L, 16);
        if (t_redl_m > 0 && (uint32_t)t_redl_m <= p.B) p.L = (uint32_t)t_redl_m;
    }
    p.T = p.B / p.L;
    p.nblk = (p.T + 255) / 256;
    // low bits sorted by the SECOND pass (its bins; the first pass splits by the bits above them).  FK_MSM_LB = 10 .. 12.
    static const int t_lb = std::min(12, std::max(10, tune("FK_MSM_LB", 10)));
    p.LB = (c - 1) < (uint32_t)t_lb ? (c - 1) : (uint32_t)t_lb;
    while ((p.B >> p.LB) > S2_MAX_HI) p.LB++;          // the first pass's LDS tables hold S2_MAX_HI bins
    p.nlo = 1u << p.LB;
    p.nhi = p.B >> p.LB;
    return p;
}

// ------------------------------------------------------------------------------------------ digits
// Signed digits over W windows that split the 255 bits (254-bit scalar + recoding carry) as evenly as possible: the
// first `wide` windows are cb + 1 bits, the rest cb bits.  A uniform width would leave a short top window (14 bits at
// c = 20) whose few buckets each receive thousands of entries and all have to go through the oversized-bucket path.
// Two scalars per lane, both loaded before either is processed: the kernel usually runs underneath another lane's
// accumulation with a single wave per SIMD, where bytes in flight per wave are what its speed depends on.
__global__ __launch_bounds__(256) void msm_digits_kernel(const Fr *scalars, size_t n, uint32_t cb, uint32_t wide, uint32_t W, uint32_t *digits) {
    const size_t half = (n + 1) / 2;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i0 >= half) return;
    const size_t i1 = i0 + half;
    const bool has1 = i1 < n;
    Fr in[2];
    in[0] = scalars[i0];
    in[1] = scalars[has1 ? i1 : i0];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (u == 1 && !has1) break;
        const size_t i = u ? i1 : i0;
        Fr s = Fr::from_mont(in[u]);   // canonical integer, 254 bits
        uint32_t carry = 0;
        for (uint32_t w = 0; w < W; w++) {
            const uint32_t cw = cb + (w < wide ? 1 : 0);
            const uint32_t o = w * cb + (w < wide ? w : wide), limb = o >> 5, sh = o & 31;
            const uint32_t B = 1u << (cw - 1), mask = (1u << cw) - 1;
            uint32_t lo = limb < 8 ? s.v[limb] : 0, hi = limb + 1 < 8 ? s.v[limb + 1] : 0;
            uint32_t raw = (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) & mask;
            raw += carry;
            uint32_t d, neg;
            if (raw > B) { d = (1u << cw) - raw; neg = 1; carry = 1; } else { d = raw; neg = 0; carry = 0; }
            digits[(size_t)w * n + i] = d | (neg << 31);
        }
    }
}

// ------------------------------------------------------------------------------------------ bucket sort
struct OverEntry { uint32_t g, size; };

// Exclusive scan over the 1024 lanes of a workgroup, one value per lane: wave64 shuffles inside a wave, the 16 wave sums through
// LDS -- two barriers instead of the twenty of a ten-step Hillis-Steele scan (the scatter kernels scan once per 16 K-entry tile).
// wsum: 16 words of LDS.  Returns the exclusive prefix; *total = sum over the workgroup.
static __device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *wsum, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= (uint32_t)off) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) { const uint32_t x = wsum[w]; all += x; if (w < wv) base += x; }
    __syncthreads();                 // wsum may be reused by the caller's next scan
    *total = all;
    return base + incl - v;
}

// ------------------------------------------------------------------------------------------ oversized buckets: device-side state
// Everything a multiplication decides from its data -- the cap a bucket-lane walks to, the list of buckets beyond it, their
// segment tasks -- is decided ON THE DEVICE and read by the following kernels from here, so that a whole multiplication is
// queued without a single host round trip (round 1 waited twice per multiplication, and every wait kept the host from queueing
// the NEXT multiplication's kernels).  One per lane; lives as long as the lane's sort (B2 reuses B1's).
struct MsmDyn {
    uint32_t cap, n_over, seg, n_tasks, n_obs, error, pad0, pad1;
    uint32_t hist[16];              // buckets with cap0 << k < size (<= cap0 << (k + 1) for k < 15), from s2_prefix2_kernel
    unsigned long long adds;        // mixed additions of the accumulate kernel (statistics)
};
static constexpr uint32_t OVER_MAX = 4096;      // oversized buckets a multiplication can table (the cap rule leaves at most ~3072)

// Appends an entry to the oversized-bucket list for every lane with `take`: ONE global atomic per wave reserves the range.
static __device__ __forceinline__ void over_append(bool take, uint32_t g, uint32_t size, OverEntry *over, MsmDyn *dyn) {
    const unsigned long long m = __ballot(take);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63, leader = (uint32_t)__ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&dyn->n_over, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (take) {
        const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (k < OVER_MAX) { over[k].g = g; over[k].size = size; } else dyn->error = 1;
    }
}

// ------------------------------------------------------------------------------------------ two-pass radix sort
// Pass 1 partitions each window's entries by the high bits of the bucket index (<= 2^11 bins, LDS counters and
// cursors), pass 2 sorts every high-bin segment by the low bits (<= 2^10 bins) in tiles of S2_TILE entries.  Both
// passes write through a few hundred open lines per workgroup, so stores combine in L2 instead of the one-pass
// scatter's 4-byte writes over 2^15 open lines (5.8x write amplification measured).
__global__ __launch_bounds__(SORT_THREADS) void s2_hist1_kernel(const uint32_t *digits, size_t n, size_t chunk, uint32_t nchunks,
                                                                 uint32_t LB, uint32_t nhi, uint32_t *cnt1) {
    extern __shared__ uint32_t hist[];
    const uint32_t ch = blockIdx.x, w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < nhi; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const size_t lo = (size_t)ch * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t *dg = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)blockDim.x * 8) {     // eight independent loads in flight per lane
        uint32_t d[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const size_t k = i + (size_t)u * blockDim.x; d[u] = k < hi ? (dg[k] & 0x7fffffffu) : 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (d[u]) atomicAdd(&hist[(d[u] - 1) >> LB], 1u);
    }
    __syncthreads();
    uint32_t *out = cnt1 + ((size_t)w * nchunks + ch) * nhi;
    for (uint32_t b = threadIdx.x; b < nhi; b += blockDim.x) out[b] = hist[b];
}

// one block per window: cnt1 -> exclusive prefix over chunks; seg_size / seg_start (relative to the window) per high bin
__global__ __launch_bounds__(1024) void s2_prefix1_kernel(uint32_t *cnt1, uint32_t nchunks, uint32_t nhi, uint32_t *seg_size, uint32_t *seg_start,
                                                           uint32_t *seg_tiles) {
    extern __shared__ uint32_t ssz[];
    const uint32_t w = blockIdx.x;
    for (uint32_t h = threadIdx.x; h < nhi; h += blockDim.x) {
        uint32_t run = 0;
        for (uint32_t ch = 0; ch < nchunks; ch++) {
            uint32_t *p = cnt1 + ((size_t)w * nchunks + ch) * nhi + h;
            const uint32_t v = *p; *p = run; run += v;
        }
        ssz[h] = run;
    }
    __syncthreads();
    // exclusive scan of the nhi (<= 2048) segment sizes: two values per lane, Hillis-Steele over the lane sums
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t v0 = 2 * t < nhi ? ssz[2 * t] : 0, v1 = 2 * t + 1 < nhi ? ssz[2 * t + 1] : 0;
    part[t] = v0 + v1;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t x = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    const uint32_t base = part[t] - (v0 + v1);
    if (2 * t < nhi) { const size_t g = (size_t)w * nhi + 2 * t; seg_size[g] = v0; seg_start[g] = base; seg_tiles[g] = (v0 + S2_TILE - 1) / S2_TILE; }
    if (2 * t + 1 < nhi) { const size_t g = (size_t)w * nhi + 2 * t + 1; seg_size[g] = v1; seg_start[g] = base + v0; seg_tiles[g] = (v1 + S2_TILE - 1) / S2_TILE; }
}

// exclusive prefix of the per-segment tile counts (<= 22 * 2^11 values); tile_start[nseg] = total.  One 1024-lane
// workgroup: each lane owns a run of consecutive segments, the lane sums are scanned in LDS.
__global__ __launch_bounds__(1024) void s2_tile_prefix_kernel(const uint32_t *seg_tiles, uint32_t nseg, uint32_t *tile_start) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (nseg + 1023) / 1024;
    const uint32_t lo = t * per, hi = lo + per < nseg ? lo + per : nseg;
    uint32_t sum = 0;
    for (uint32_t s = lo; s < hi; s++) sum += seg_tiles[s];
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t s = lo; s < hi; s++) { tile_start[s] = run; run += seg_tiles[s]; }
    if (t == 1023) tile_start[nseg] = part[1023];
}

// First-pass scatter, LDS-staged like the second one: the chunk is walked in sub-tiles of S1_TILE entries; each sub-tile is
// counted per high bin, laid out in bin order in LDS and written as contiguous runs (idx: 4 B, low bits: 2 B per entry) behind
// the workgroup's per-bin cursors -- s2_scatter1n_body below (round 1's register-ranked form needed 128 registers and spilled).
// which segment does tile `t` belong to (tile_start is non-decreasing; empty segments own no tile)
static __device__ __forceinline__ uint32_t s2_find_segment(const uint32_t *tile_start, uint32_t nseg, uint32_t t) {
    uint32_t lo = 0, hi = nseg;          // invariant: tile_start[lo] <= t < tile_start[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile_start[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void s2_hist2_kernel(const uint16_t *tmp_lo, size_t n, uint32_t nhi, uint32_t nlo, const uint32_t *tile_start,
                                                        uint32_t nseg, const uint32_t *seg_start, const uint32_t *seg_size, uint32_t *cnt2) {
    extern __shared__ uint32_t hist[];
    const uint32_t tile = blockIdx.x;
    if (tile >= tile_start[nseg]) return;        // the grid is the host's upper bound on the tile count (no read-back)
    const uint32_t sgm = s2_find_segment(tile_start, nseg, tile), w = sgm / nhi, t = tile - tile_start[sgm];
    for (uint32_t b = threadIdx.x; b < nlo; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t size = seg_size[sgm], lo = t * S2_TILE, hi = lo + S2_TILE < size ? lo + S2_TILE : size;
    const uint16_t *src = tmp_lo + (size_t)w * n + seg_start[sgm];
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 256 * 8) {        // eight independent loads in flight per lane
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { const uint32_t kk = k + u * 256; v[u] = kk < hi ? src[kk] : 0xffffffffu; }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) if (v[u] != 0xffffffffu) atomicAdd(&hist[v[u]], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nlo; b += 256) cnt2[(size_t)tile * nlo + b] = hist[b];
}

// one WAVE per high-bin segment (a 1024-lane workgroup per segment spent its time being launched: 24 576 of them per
// multiplication at 2^25, 1.9 ms): lane l owns the low bins l, l + 64, ... (up to 64 of them, in registers); per tile all of a
// lane's counters are loaded at once (independent loads) and replaced by the running prefix over the segment's tiles; then a
// wave scan per 64-bin chunk gives the bucket starts.  Also bucket totals and the size histogram above the planned cap
// (msm_cap_kernel).
static constexpr uint32_t P2_MAX = 32;      // low bins per lane and wave: nlo <= 2048 per wave pass (larger nlo: two passes over the tiles)
__global__ __launch_bounds__(256) void s2_prefix2_kernel(uint32_t *cnt2, uint32_t nseg, uint32_t nhi, uint32_t nlo, uint32_t B, const uint32_t *tile_start,
                                                          const uint32_t *seg_start, uint32_t cap0, uint32_t *totals, uint32_t *starts, MsmDyn *dyn) {
    __shared__ uint32_t sh_hist[16];
    const uint32_t lane = threadIdx.x & 63, sgm = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (threadIdx.x < 16) sh_hist[threadIdx.x] = 0;
    __syncthreads();
    if (sgm < nseg) {
        const uint32_t w = sgm / nhi, h = sgm % nhi, t0 = tile_start[sgm], t1 = tile_start[sgm + 1];
        uint32_t carry = seg_start[sgm];
        for (uint32_t b0 = 0; b0 < nlo; b0 += 64 * P2_MAX) {          // P2_MAX 64-bin chunks at a time
            const uint32_t per = (nlo - b0 + 63) / 64 < P2_MAX ? (nlo - b0 + 63) / 64 : P2_MAX;
            uint32_t run[P2_MAX];
#pragma unroll
            for (uint32_t i = 0; i < P2_MAX; i++) run[i] = 0;
            for (uint32_t t = t0; t < t1; t++) {
                uint32_t *row = cnt2 + (size_t)t * nlo + b0;
                uint32_t v[P2_MAX];
#pragma unroll
                for (uint32_t i = 0; i < P2_MAX; i++) { const uint32_t b = i * 64 + lane; v[i] = (i < per && b0 + b < nlo) ? row[b] : 0; }
#pragma unroll
                for (uint32_t i = 0; i < P2_MAX; i++) { const uint32_t b = i * 64 + lane; if (i < per && b0 + b < nlo) row[b] = run[i]; run[i] += v[i]; }
            }
#pragma unroll
            for (uint32_t i = 0; i < P2_MAX; i++) {
                if (i < per) {          // uniform over the wave
                    const uint32_t b = b0 + i * 64 + lane;
                    uint32_t incl = run[i];
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= (uint32_t)off) incl += x; }
                    if (b < nlo) {
                        const size_t g = (size_t)w * B + (size_t)h * nlo + b;
                        totals[g] = run[i];
                        starts[g] = carry + incl - run[i];
                        if (run[i] > cap0) {        // how far over the statistical cap: class k = the largest k with run > cap0 << k (msm_cap_kernel)
                            const uint32_t k = 31u - (uint32_t)__clz((run[i] - 1) / cap0);
                            atomicAdd(&sh_hist[k < 15 ? k : 15], 1u);
                        }
                    }
                    carry += (uint32_t)__shfl((int)incl, 63, 64);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 16 && sh_hist[threadIdx.x]) atomicAdd(&dyn->hist[threadIdx.x], sh_hist[threadIdx.x]);
}

// The cap assumes Poisson bucket loads.  Scalars with many repeated values (a batch witness) put thousands of buckets a little
// over it, and a wave per segment plus a 256-lane fold workgroup per such bucket is a poor trade: the cap is doubled until at
// most `many` buckets remain above it (or it reaches 2^20).  One lane; resets the rest of the state for this sort.
__global__ void msm_cap_kernel(MsmDyn *dyn, uint32_t cap0, uint32_t many) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t above[17]; above[16] = 0;
    for (int k = 15; k >= 0; k--) above[k] = above[k + 1] + dyn->hist[k];
    uint32_t k = 0;
    while (k < 15 && above[k] > many && ((unsigned long long)cap0 << k) < (1ull << 20)) k++;
    const unsigned long long c = (unsigned long long)cap0 << k;
    dyn->cap = c < (1ull << 30) ? (uint32_t)c : (1u << 30);
    dyn->n_over = 0; dyn->seg = 0; dyn->n_tasks = 0; dyn->n_obs = 0; dyn->error = 0; dyn->adds = 0;
}

// Lists the buckets longer than the (final) cap.
__global__ __launch_bounds__(256) void msm_over_list_kernel(const uint32_t *totals, size_t WB, OverEntry *over, MsmDyn *dyn) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sz = g < WB ? totals[g] : 0;
    over_append(g < WB && sz > dyn->cap, (uint32_t)g, sz, over, dyn);
}

// Second-pass scatter with LDS staging: entries of the tile are ranked per low bin (LDS counters), placed in bin order in
// an LDS staging buffer and then written out so that consecutive lanes store consecutive addresses of one bucket run
// (64-byte runs on average) instead of one 4-byte store per lane to an arbitrary line.
template <uint32_t MAXLO>      // low bins the LDS tables are sized for (1024, 2048, 4096)
__global__ __launch_bounds__(1024) void s2_scatter2_kernel(const uint32_t *tmp_idx, const uint16_t *tmp_lo, size_t n, uint32_t nhi, uint32_t nlo, uint32_t B,
                                                            const uint32_t *tile_start, uint32_t nseg, const uint32_t *seg_start, const uint32_t *seg_size,
                                                            const uint32_t *cnt2, const uint32_t *starts, uint32_t *sorted) {
    constexpr uint32_t BPL = MAXLO / 1024;   // bins per lane
    __shared__ uint32_t lcnt[MAXLO];         // per-bin count, then exclusive offset inside the tile
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t gbase[MAXLO];        // global position (window-relative) of this tile's first entry of the bin
    __shared__ uint32_t stage_idx[S2_TILE];
    __shared__ uint16_t stage_lo[S2_TILE];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= tile_start[nseg]) return;        // see s2_hist2_kernel
    const uint32_t sgm = s2_find_segment(tile_start, nseg, tile), w = sgm / nhi, h = sgm % nhi, t = tile - tile_start[sgm];
#pragma unroll
    for (uint32_t i = 0; i < BPL; i++) {
        const uint32_t b = tid + i * 1024;
        lcnt[b] = 0;
        gbase[b] = b < nlo ? starts[(size_t)w * B + (size_t)h * nlo + b] + cnt2[(size_t)tile * nlo + b] : 0;
    }
    __syncthreads();
    const uint32_t size = seg_size[sgm], lo = t * S2_TILE, cnt = (lo + S2_TILE < size ? lo + S2_TILE : size) - lo;
    const size_t base = (size_t)w * n + seg_start[sgm] + lo;
    uint32_t e_idx[S2_EPT], e_lo[S2_EPT], e_rank[S2_EPT];
#pragma unroll
    for (uint32_t j = 0; j < S2_EPT; j++) {
        const uint32_t k = tid + j * 1024;
        if (k < cnt) { e_idx[j] = tmp_idx[base + k]; e_lo[j] = tmp_lo[base + k]; e_rank[j] = atomicAdd(&lcnt[e_lo[j]], 1u); }
    }
    __syncthreads();
    // exclusive scan of the bin counts: BPL consecutive bins per lane
    uint32_t c[BPL], mine = 0;
#pragma unroll
    for (uint32_t i = 0; i < BPL; i++) { c[i] = lcnt[tid * BPL + i]; mine += c[i]; }
    uint32_t all_;
    uint32_t excl = block_excl_scan_1024(mine, wsum, &all_);
#pragma unroll
    for (uint32_t i = 0; i < BPL; i++) { lcnt[tid * BPL + i] = excl; excl += c[i]; }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < S2_EPT; j++) {
        const uint32_t k = tid + j * 1024;
        if (k < cnt) { const uint32_t q = lcnt[e_lo[j]] + e_rank[j]; stage_idx[q] = e_idx[j]; stage_lo[q] = (uint16_t)e_lo[j]; }
    }
    __syncthreads();
    uint32_t *out = sorted + (size_t)w * n;
    for (uint32_t q = tid; q < cnt; q += 1024) {
        const uint32_t b = stage_lo[q];
        out[gbase[b] + (q - lcnt[b])] = stage_idx[q];
    }
}

// First-pass scatter (two-atomic form): a lane owns S1_TILE / NT entries and keeps nothing about them between the phases -- the
// counting phase only counts, the placing phase draws each entry's slot from a per-bin LDS cursor (round 1's register-ranked
// form needed 128 registers and spilled).  256- and 512-lane shapes of both scatter kernels ("thin" workgroups that fit where one
// accumulate workgroup has left) were built and measured in rounds 1 and 2: slower alone and no better underneath -- removed.
template <uint32_t NT>
static __device__ __forceinline__ uint32_t block_excl_scan_nt(uint32_t v, uint32_t *wsum, uint32_t *total) {
    constexpr uint32_t NW = NT / 64;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= (uint32_t)off) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) { const uint32_t x = wsum[w]; all += x; if (w < wv) base += x; }
    __syncthreads();
    *total = all;
    return base + incl - v;
}

template <uint32_t NT>
static __device__ __forceinline__ void s2_scatter1n_body(const uint32_t *digits, size_t n, size_t chunk, uint32_t nchunks, uint32_t LB,
                                                           uint32_t nhi, const uint32_t *cnt1, const uint32_t *seg_start, uint32_t *tmp_idx,
                                                           uint16_t *tmp_lo) {
    constexpr uint32_t BPL = S2_MAX_HI / NT;        // bins per lane
    constexpr uint32_t EPT = S1_TILE / NT;          // entries per lane and sub-tile
#ifdef FK_S1_NO_PREFETCH
    constexpr bool PREFETCH = false;
#else
    constexpr bool PREFETCH = EPT <= 16;            // the sub-tile's digits live in registers and the NEXT sub-tile's are loaded while this one is written out
#endif
    __shared__ uint32_t cursor[S2_MAX_HI];
    __shared__ uint32_t lcnt[S2_MAX_HI];
    __shared__ uint32_t lexc[S2_MAX_HI];        // exclusive offsets inside the sub-tile; advanced to the bins' ends by the placing phase
    __shared__ uint32_t part[16];
    __shared__ uint32_t stage_idx[S1_TILE];
    __shared__ uint16_t stage_lo[S1_TILE];
    __shared__ uint16_t stage_bin[S1_TILE];
    const uint32_t ch = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const uint32_t *cnt = cnt1 + ((size_t)w * nchunks + ch) * nhi;
    for (uint32_t b = tid; b < S2_MAX_HI; b += NT) cursor[b] = b < nhi ? seg_start[(size_t)w * nhi + b] + cnt[b] : 0;
    const size_t c_lo = (size_t)ch * chunk, c_hi = c_lo + chunk < n ? c_lo + chunk : n;
    const uint32_t *dg = digits + (size_t)w * n;
    const uint32_t lomask = (1u << LB) - 1;
    uint32_t *oidx = tmp_idx + (size_t)w * n;
    uint16_t *olo = tmp_lo + (size_t)w * n;
    uint32_t d[PREFETCH ? EPT : 1], dn[PREFETCH ? EPT : 1];
    if (PREFETCH && c_lo < c_hi) {
        const uint32_t cnt0 = (uint32_t)(c_hi - c_lo < S1_TILE ? c_hi - c_lo : S1_TILE);
#pragma unroll
        for (uint32_t j = 0; j < EPT; j++) { const uint32_t k = tid + j * NT; d[j] = k < cnt0 ? dg[c_lo + k] : 0; }
    }
    for (size_t sub = c_lo; sub < c_hi; sub += S1_TILE) {
        const uint32_t cntt = (uint32_t)(c_hi - sub < S1_TILE ? c_hi - sub : S1_TILE);
        const uint32_t *src = dg + sub;
        for (uint32_t b = tid; b < S2_MAX_HI; b += NT) lcnt[b] = 0;
        __syncthreads();
        if (PREFETCH) {
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) { const uint32_t bkt = d[j] & 0x7fffffffu; if (bkt) atomicAdd(&lcnt[(bkt - 1) >> LB], 1u); }     // (a digit past the end was loaded as 0)
        } else {
#pragma unroll 8
            for (uint32_t k = tid; k < cntt; k += NT) {
                const uint32_t bkt = src[k] & 0x7fffffffu;
                if (bkt) atomicAdd(&lcnt[(bkt - 1) >> LB], 1u);
            }
        }
        __syncthreads();
        uint32_t c[BPL], mine = 0;
#pragma unroll
        for (uint32_t i = 0; i < BPL; i++) { c[i] = lcnt[tid * BPL + i]; mine += c[i]; }
        uint32_t total;
        uint32_t ex = block_excl_scan_nt<NT>(mine, part, &total);
#pragma unroll
        for (uint32_t i = 0; i < BPL; i++) { lexc[tid * BPL + i] = ex; ex += c[i]; }
        __syncthreads();
        if (PREFETCH) {
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) {
                const uint32_t bkt = d[j] & 0x7fffffffu;
                if (bkt) {
                    const uint32_t bin = (bkt - 1) >> LB, q = atomicAdd(&lexc[bin], 1u);
                    stage_idx[q] = (uint32_t)(sub + tid + j * NT) | (d[j] & 0x80000000u); stage_lo[q] = (uint16_t)((bkt - 1) & lomask); stage_bin[q] = (uint16_t)bin;
                }
            }
            // the next sub-tile's digits: in flight while this one is written out
            const size_t nxt = sub + S1_TILE;
            const uint32_t cntn = nxt < c_hi ? (uint32_t)(c_hi - nxt < S1_TILE ? c_hi - nxt : S1_TILE) : 0;
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) { const uint32_t k = tid + j * NT; dn[j] = k < cntn ? dg[nxt + k] : 0; }
        } else {
#pragma unroll 8
            for (uint32_t k = tid; k < cntt; k += NT) {
                const uint32_t dd = src[k], bkt = dd & 0x7fffffffu;
                if (bkt) {
                    const uint32_t bin = (bkt - 1) >> LB, q = atomicAdd(&lexc[bin], 1u);
                    stage_idx[q] = (uint32_t)(sub + k) | (dd & 0x80000000u); stage_lo[q] = (uint16_t)((bkt - 1) & lomask); stage_bin[q] = (uint16_t)bin;
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t q = tid; q < total; q += NT) {
            const uint32_t bn = stage_bin[q];
            const uint32_t dst = cursor[bn] + (q - (lexc[bn] - lcnt[bn]));
            oidx[dst] = stage_idx[q];
            olo[dst] = stage_lo[q];
        }
        __syncthreads();
        for (uint32_t b = tid; b < S2_MAX_HI; b += NT) cursor[b] += lcnt[b];
        if (PREFETCH) {
#pragma unroll
            for (uint32_t j = 0; j < EPT; j++) d[j] = dn[j];
        }
        __syncthreads();
    }
}

// The same pass on a register diet (experiment builds, -DFK_S1_LEAN with -DFK_S1_NT=256): 32 VGPRs, so that one wave of it fits
// into the 32 registers per SIMD lane that the G2 accumulation's two 240-register waves leave (profiles/r05_sort_kernel_resources.txt) and the pass
// can run BESIDE that accumulation instead of waiting for drained compute units.  What the diet costs: no unrolling of the global loops (one load in
// flight per lane, FK_S1_LEAN_VEC=4: one 16-byte load), 32-bit offsets inside the chunk, scheduling barriers that keep the compiler from software-
// pipelining the write-out loop (which alone took the allocation from 32 to 46).  Same arguments, same result as s2_scatter1n_body.
#ifdef FK_S1_LEAN
#ifndef FK_S1_LEAN_TILE
#define FK_S1_LEAN_TILE 4096
#endif
static constexpr uint32_t LEAN_TILE = FK_S1_LEAN_TILE;      // its own sub-tile: the production kernel of the same build keeps S1_TILE
template <uint32_t NT>
static __device__ __forceinline__ void s2_scatter1_lean_body(const uint32_t *digits, size_t n, size_t chunk, uint32_t nchunks, uint32_t LB,
                                                              uint32_t nhi, const uint32_t *cnt1, const uint32_t *seg_start, uint32_t *tmp_idx,
                                                              uint16_t *tmp_lo) {
    constexpr uint32_t BPL = S2_MAX_HI / NT;
    __shared__ uint32_t cursor[S2_MAX_HI];
    __shared__ uint32_t lcnt[S2_MAX_HI];
    __shared__ uint32_t lexc[S2_MAX_HI];
    __shared__ uint32_t part[16];
    __shared__ uint32_t stage_idx[LEAN_TILE];
    __shared__ uint16_t stage_lo[LEAN_TILE];
    __shared__ uint16_t stage_bin[LEAN_TILE];
    __shared__ uint32_t stage_dig[LEAN_TILE];       // the sub-tile's digits: global memory is read once per entry
    const uint32_t ch = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const uint32_t *cnt = cnt1 + ((size_t)w * nchunks + ch) * nhi;
    for (uint32_t b = tid; b < S2_MAX_HI; b += NT) cursor[b] = b < nhi ? seg_start[(size_t)w * nhi + b] + cnt[b] : 0;
    const size_t c_lo = (size_t)ch * chunk, c_hi = c_lo + chunk < n ? c_lo + chunk : n;
    const uint32_t lomask = (1u << LB) - 1;
    uint32_t *oidx = tmp_idx + (size_t)w * n;
    uint16_t *olo = tmp_lo + (size_t)w * n;
    const uint32_t clen = (uint32_t)(c_hi > c_lo ? c_hi - c_lo : 0);
    const uint32_t *dgc = digits + (size_t)w * n + c_lo;
    const uint32_t base_idx = (uint32_t)c_lo;
    for (uint32_t so = 0; so < clen; so += LEAN_TILE) {
        const uint32_t cntt = clen - so < LEAN_TILE ? clen - so : LEAN_TILE;
        const uint32_t *src = dgc + so;
        const uint32_t sub = base_idx + so;
        for (uint32_t b = tid; b < S2_MAX_HI; b += NT) lcnt[b] = 0;
        __syncthreads();
        // counting: the digits are parked in LDS (stage_idx) on the way, so global memory is read ONCE per entry
#pragma clang loop unroll(disable)
        for (uint32_t k = tid; k < cntt; k += NT) {
            const uint32_t dd = src[k];
            stage_dig[k] = dd;
            const uint32_t bkt = dd & 0x7fffffffu;
            if (bkt) atomicAdd(&lcnt[(bkt - 1) >> LB], 1u);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        uint32_t c[BPL], mine = 0;
#pragma unroll
        for (uint32_t i = 0; i < BPL; i++) { c[i] = lcnt[tid * BPL + i]; mine += c[i]; }
        uint32_t total;
        uint32_t ex = block_excl_scan_nt<NT>(mine, part, &total);
#pragma unroll
        for (uint32_t i = 0; i < BPL; i++) { lexc[tid * BPL + i] = ex; ex += c[i]; }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // placing: the digits come back out of LDS (stage_dig), each entry draws its slot from its bin's cursor
#pragma clang loop unroll(disable)
        for (uint32_t k = tid; k < cntt; k += NT) {
            const uint32_t dd = stage_dig[k], bkt = dd & 0x7fffffffu;
            if (bkt) {
                const uint32_t bin = (bkt - 1) >> LB, q = atomicAdd(&lexc[bin], 1u);
                stage_idx[q] = (sub + k) | (dd & 0x80000000u); stage_lo[q] = (uint16_t)((bkt - 1) & lomask); stage_bin[q] = (uint16_t)bin;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
        for (uint32_t q = tid; q < total; q += NT) {
            const uint32_t bn = stage_bin[q];
            const uint32_t dst = cursor[bn] + (q - (lexc[bn] - lcnt[bn]));
            oidx[dst] = stage_idx[q];
            olo[dst] = stage_lo[q];
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        for (uint32_t b = tid; b < S2_MAX_HI; b += NT) cursor[b] += lcnt[b];
        __syncthreads();
    }
}
#endif

#define S2N_ARGS1 const uint32_t *digits, size_t n, size_t chunk, uint32_t nchunks, uint32_t LB, uint32_t nhi, const uint32_t *cnt1, const uint32_t *seg_start, uint32_t *tmp_idx, uint16_t *tmp_lo
#define S2N_PASS1 digits, n, chunk, nchunks, LB, nhi, cnt1, seg_start, tmp_idx, tmp_lo
__global__ __launch_bounds__(1024) void s2_scatter1_n1024_kernel(S2N_ARGS1) { s2_scatter1n_body<1024>(S2N_PASS1); }
// experiment builds only (-DFK_S1_NT=256 -DFK_S1_TILE=4096 [-DFK_S1_NO_PREFETCH]): the same pass in thin workgroups -- one wave per SIMD and few registers,
// which could sit BESIDE the G2 accumulation's two 226-register waves per SIMD (rounds 1-2 measured thin shapes underneath G1 accumulations only)
#ifdef FK_S1_NT
#ifdef FK_S1_LEAN
__global__ __launch_bounds__(FK_S1_NT) void s2_scatter1_thin_kernel(S2N_ARGS1) { s2_scatter1_lean_body<FK_S1_NT>(S2N_PASS1); }
#else
__global__ __launch_bounds__(FK_S1_NT) void s2_scatter1_thin_kernel(S2N_ARGS1) { s2_scatter1n_body<FK_S1_NT>(S2N_PASS1); }
#endif
#endif

// ------------------------------------------------------------------------------------------ bucket -> lane assignment
// A wave runs as long as its longest bucket, so lanes are handed buckets of (nearly) equal length: buckets are
// binned by min(size, cap) into 1024 classes, largest first (counting sort on the class), and lane t of the
// accumulate kernel takes bucket perm[t].  VALUUtilization of the accumulate kernels was 80-86 % without this.
static constexpr uint32_t SIZE_BINS = 1024;
// monotone, largest first: sizes below 768 get one class each (classes 256..1023), larger ones share 256 coarse classes
static __device__ __forceinline__ uint32_t size_class(uint32_t size, uint32_t cap) {
    const uint32_t s = size < cap ? size : cap;
    if (s < 768) return (SIZE_BINS - 1) - s;
    const uint32_t span = cap > 768 ? cap - 768 : 1;
    return 255u - (uint32_t)(((uint64_t)(s - 768) * 255u) / span);
}
// wave64 sum, result in lane 0
static __device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += (unsigned long long)__shfl_down((long long)v, off, 64);
    return v;
}
// adds (may be null): += sum over the buckets of min(size, cap) - 1, i.e. the mixed additions the accumulate kernel will do
// (a bucket's first entry is a copy) -- the unit of the VALU roofline in bench.py.  `count_adds`: off for the merged form,
// whose buckets span the windows (msm_merge_totals_kernel counts there).
static __device__ __forceinline__ uint32_t dyn_cap(const MsmDyn *dyn, uint32_t mult) {      // cap (x W for merged bucket lengths)
    const unsigned long long c = (unsigned long long)dyn->cap * mult;
    return c < (1ull << 30) ? (uint32_t)c : (1u << 30);
}
__global__ __launch_bounds__(256) void msm_size_hist_kernel(const uint32_t *totals, size_t WB, const MsmDyn *dyn, uint32_t mult, uint32_t *bins, unsigned long long *adds) {
    __shared__ uint32_t sh[SIZE_BINS];
    const uint32_t cap = dyn_cap(dyn, mult);
    for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256) sh[i] = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < WB; g += (size_t)gridDim.x * 256) {
        const uint32_t t = totals[g];
        atomicAdd(&sh[size_class(t, cap)], 1u);
        if (t) mine += (t < cap ? t : cap) - 1;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256) if (sh[i]) atomicAdd(&bins[i], sh[i]);
    if (adds) { mine = wave_sum_u64(mine); if ((threadIdx.x & 63) == 0 && mine) atomicAdd(adds, mine); }
}
__global__ __launch_bounds__(SIZE_BINS) void msm_size_scan_kernel(uint32_t *bins) {      // exclusive scan in place
    __shared__ uint32_t sh[SIZE_BINS];
    const uint32_t t = threadIdx.x, v = bins[t];
    sh[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < SIZE_BINS; off <<= 1) {
        uint32_t x = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    bins[t] = sh[t] - v;
}
// ranks inside the workgroup come from LDS counters; one global atomic per (workgroup, non-empty class) reserves the
// range -- with millions of similar-sized buckets a per-bucket global atomic serialises on a few dozen hot cursors
__global__ __launch_bounds__(1024) void msm_size_scatter_kernel(const uint32_t *totals, size_t WB, const MsmDyn *dyn, uint32_t mult, uint32_t *cursor, uint32_t *perm) {
    __shared__ uint32_t cnt[SIZE_BINS];
    __shared__ uint32_t base[SIZE_BINS];
    const uint32_t cap = dyn_cap(dyn, mult);
    const size_t g = (size_t)blockIdx.x * 1024 + threadIdx.x;
    cnt[threadIdx.x] = 0;                       // SIZE_BINS == blockDim.x == 1024
    __syncthreads();
    uint32_t cls = 0, rank = 0;
    if (g < WB) { cls = size_class(totals[g], cap); rank = atomicAdd(&cnt[cls], 1u); }
    __syncthreads();
    if (cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
    if (g < WB) perm[base[cls] + rank] = (uint32_t)g;
}

// ------------------------------------------------------------------------------------------ wave / block reductions
template <class T>
static __device__ __forceinline__ T shfl_down_obj(const T &v, int off) {
    static_assert(sizeof(T) % 4 == 0, "word sized");
    T r;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(&v);
    uint32_t *d = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) d[i] = (uint32_t)__shfl_down((int)s[i], off, 64);
    return r;
}

// wave64 tree: after the call lane 0 holds the sum of all 64 lanes' points
template <class F>
static __device__ __forceinline__ void wave_reduce(Xyzz<F> &acc) {
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
        Xyzz<F> o = shfl_down_obj(acc, off);
        acc.add(o);
    }
}

// 256-thread block: result valid in thread 0.  `sh` holds 4 points.
template <class F>
static __device__ __forceinline__ void block_reduce_256(Xyzz<F> &acc, Xyzz<F> *sh) {
    wave_reduce(acc);
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = sh[0];
        for (int k = 1; k < 4; k++) { Xyzz<F> o = sh[k]; acc.add(o); }
    }
}

// ------------------------------------------------------------------------------------------ bucket accumulation
// The per-lane walk of a bucket: the 8 x 32-bit XYZZ accumulator of curve.hpp.  (A 9 x 29-bit-limb accumulator -- no carry-out
// per multiply-accumulate, no conditional subtractions -- was built in round 2, bit-exact and NOT faster in place: 234 against
// 216 ms per proof; DESIGN.md section 3.3.  Removed in round 3.)
#if defined(__HIP_DEVICE_COMPILE__)
template <class F, int MODE> struct Walker {       // MODE 0
    Xyzz<F> acc = Xyzz<F>::inf();
    __device__ __forceinline__ void add(const Affine<F> &p, bool neg) { acc.add_mixed(affine_neg_if(p, neg)); }
    __device__ __forceinline__ Xyzz<F> result() const { return acc; }
};
// MODE 2: the accumulator's coordinates live in [0, 2p) (field.hpp, LAZY): no conditional subtraction behind any of the ten
// products of a mixed addition; the bucket is made canonical when it is stored.  A loaded point is canonical, hence valid.
template <class F> struct Walker<F, 2> {
    using FL = typename LazyOf<F>::type;
    Xyzz<FL> acc = Xyzz<FL>::inf();
    __device__ __forceinline__ void add(const Affine<F> &p, bool neg) {
        if (p.is_inf()) return;                 // tested on the canonical form (one compare chain instead of two)
        static_assert(sizeof(Affine<FL>) == sizeof(Affine<F>), "layout");
        Affine<FL> q;
        __builtin_memcpy(&q, &p, sizeof q);
        if (neg) q.y = FL::neg(q.y);
        acc.add_mixed_nz(q);
    }
    __device__ __forceinline__ Xyzz<F> result() const {
        if (acc.is_inf()) return Xyzz<F>::inf();
        return Xyzz<F>{canon(acc.x), canon(acc.y), canon(acc.zz), canon(acc.zzz)};
    }
};
#else
template <class F, int MODE> struct Walker {       // host pass: declarations only
    void add(const Affine<F> &, bool) {}
    Xyzz<F> result() const { return Xyzz<F>::inf(); }
};
#endif
// MINW = minimum waves per SIMD the register allocator must leave room for.  G1: 4 (<= 128 registers, no spills).
// G2: 2.  With the compiler's own add/sub code the inlined Fq2 mixed addition wanted 256 VGPRs + ~180 AGPRs and forcing 2
// waves spilled ~260 registers (41 ms vs 33 ms at 2^25); since the generated carry-chain add/sub (addsub_gfx950.inc) it
// needs 226 VGPRs and runs at 2 waves per SIMD without spills: 23.4 -> 13.7 ms.
template <class F, int MINW, int MODE>
__global__ __launch_bounds__(256, MINW) void msm_accumulate_kernel(const Affine<F> *bases, const uint32_t *sorted, size_t n,
                                                             const uint32_t *starts, const uint32_t *totals, uint32_t B,
                                                             uint32_t W, const MsmDyn *dyn, const uint32_t *perm, Xyzz<F> *buckets) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)W * B) return;
    const size_t g = perm[t];                 // buckets of similar length share a wave
    const uint32_t w = (uint32_t)(g / B);
    const uint32_t cap = dyn->cap;
    const uint32_t *src = sorted + (size_t)w * n + starts[g];
    uint32_t cnt = totals[g];
    if (cnt > cap) cnt = cap;
    Walker<F, MODE> acc;
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t e = src[k];
        Affine<F> p = bases[e & 0x7fffffffu];
        acc.add(p, (e >> 31) != 0);
    }
    buckets[g] = acc.result();
}

// Merged form (precomputed levels): lane t owns bucket b = perm[t] of the ONE bucket set and walks that bucket's entries in
// every window, taking window w's points from level w (2^offset_w * P) -- the weights of all windows' buckets coincide.
// The walk is ONE loop over the bucket's merged length mt[b] with a (window, position) cursor: the lanes of a wave are
// size-ordered by that merged length, so they stay in step; a loop per window would run every window to the longest of the 64
// per-window counts (Poisson: 1.6 x the mean at a load of 16).
template <class F, int MINW, int MODE>
__global__ __launch_bounds__(256, MINW) void msm_accumulate_merged_kernel(const Affine<F> *bases, const Affine<F> *lev, const uint32_t *sorted, size_t n,
                                                                    const uint32_t *starts, const uint32_t *totals, uint32_t B,
                                                                    uint32_t W, const MsmDyn *dyn, const uint32_t *perm, const uint32_t *mt, Xyzz<F> *buckets) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const uint32_t cap = dyn->cap;
    const uint32_t b = perm[t];
    const uint32_t total = mt[b];
    Walker<F, MODE> acc;
    uint32_t w = 0, k = 0, cnt = 0;
    const uint32_t *src = nullptr;
    const Affine<F> *bw = bases;
    bool first = true;
    for (uint32_t i = 0; i < total; i++) {
        while (first || k == cnt) {          // next window that holds entries of this bucket (total > i guarantees there is one)
            if (!first) w++;
            first = false;
            const size_t g = (size_t)w * B + b;
            cnt = totals[g];
            if (cnt > cap) cnt = cap;
            src = sorted + (size_t)w * n + starts[g];
            bw = w ? lev + (size_t)(w - 1) * n : bases;
            k = 0;
        }
        const uint32_t e = src[k++];
        Affine<F> p = bw[e & 0x7fffffffu];
        acc.add(p, (e >> 31) != 0);
    }
    buckets[b] = acc.result();
}
// mt[b] = sum over the windows of min(totals[w][b], cap): the merged bucket's length, for the size ordering
__global__ __launch_bounds__(256) void msm_merge_totals_kernel(const uint32_t *totals, uint32_t B, uint32_t W, const MsmDyn *dyn, uint32_t *mt, unsigned long long *adds) {
    const uint32_t cap = dyn->cap;
    unsigned long long mine = 0;       // mixed additions of this lane's buckets (a bucket's first entry is a copy)
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {       // grid-stride: ONE atomic per wave at the end (8192 atomics on one address cost 0.4 ms per launch)
        uint32_t s = 0;
        for (uint32_t w = 0; w < W; w++) { const uint32_t v = totals[(size_t)w * B + b]; s += v < cap ? v : cap; }
        mt[b] = s;
        mine += s ? s - 1 : 0;
    }
    const unsigned long long all = wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0 && all) atomicAdd(adds, all);
}
// level w+1 = 2^bits * level w, affine in, affine out.  NB (4 for G1, 2 for G2: registers) points per lane share one
// inversion (Montgomery's trick on the products zz * zzz; a point at infinity takes part with the factor 1) and their
// doubling chains run in step.  Written without arrays: a dynamically indexed array of points would live in scratch.
template <class F>
static __device__ __forceinline__ Xyzz<F> level_load(const Affine<F> *in, size_t i, size_t n) { return i < n ? Xyzz<F>::from_affine(in[i]) : Xyzz<F>::inf(); }
template <class F>
static __device__ __forceinline__ F level_den(const Xyzz<F> &p) { return p.is_inf() ? F::one() : F::mul(p.zz, p.zzz); }
template <class F>
static __device__ __forceinline__ void level_store(Affine<F> *out, size_t i, size_t n, const Xyzz<F> &p, const F &ti) {      // ti = 1 / (zz zzz)
    if (i < n) out[i] = p.is_inf() ? Affine<F>::inf() : Affine<F>{F::mul(p.x, F::mul(ti, p.zzz)), F::mul(p.y, F::mul(ti, p.zz))};
}
template <class F, int NB>
__global__ __launch_bounds__(256) void msm_level_kernel(const Affine<F> *in, size_t n, uint32_t bits, Affine<F> *out) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * NB;
    if (i0 >= n) return;
    Xyzz<F> p0 = level_load(in, i0, n), p1 = level_load(in, i0 + 1, n), p2 = Xyzz<F>::inf(), p3 = Xyzz<F>::inf();
    if (NB == 4) { p2 = level_load(in, i0 + 2, n); p3 = level_load(in, i0 + 3, n); }
    for (uint32_t k = 0; k < bits; k++) {
        p0 = Xyzz<F>::dbl(p0); p1 = Xyzz<F>::dbl(p1);
        if (NB == 4) { p2 = Xyzz<F>::dbl(p2); p3 = Xyzz<F>::dbl(p3); }
    }
    const F t0 = level_den(p0), t1 = level_den(p1);
    const F pre2 = F::mul(t0, t1);
    if (NB == 4) {
        const F t2 = level_den(p2), t3 = level_den(p3);
        const F pre3 = F::mul(pre2, t2);
        F inv = F::inv(F::mul(pre3, t3));
        level_store(out, i0 + 3, n, p3, F::mul(inv, pre3)); inv = F::mul(inv, t3);
        level_store(out, i0 + 2, n, p2, F::mul(inv, pre2)); inv = F::mul(inv, t2);
        level_store(out, i0 + 1, n, p1, F::mul(inv, t0)); inv = F::mul(inv, t1);
        level_store(out, i0, n, p0, inv);
    } else {
        F inv = F::inv(pre2);
        level_store(out, i0 + 1, n, p1, F::mul(inv, t0)); inv = F::mul(inv, t1);
        level_store(out, i0, n, p0, inv);
    }
}

struct Task { uint32_t g, seg; };
struct OverBucket { uint32_t g, task0, ntask; };

// one WAVE per SEG-entry segment of an oversized bucket (entries beyond `cap`): many small workgroups keep
// several waves per SIMD in flight (the multiply is a serial chain, one wave alone cannot fill the VALU).
// The per-lane walk uses the inlined multiply (F); the wave64 shuffle reduction runs on the
// layout-identical cold twin (FC).
template <class F, class FC, int MODE>
__global__ __launch_bounds__(64) void msm_overflow_kernel(const Affine<F> *bases0, const Affine<F> *lev, const uint32_t *sorted, size_t n,
                                                          const uint32_t *starts, const uint32_t *totals, uint32_t B,
                                                          const MsmDyn *dyn, const Task *tasks, Xyzz<FC> *partials) {
    const uint32_t cap = dyn->cap, SEG = dyn->seg, n_tasks = dyn->n_tasks;
    for (uint32_t task = blockIdx.x; task < n_tasks; task += gridDim.x) {       // the grid is fixed: the host does not know the count
        const Task t = tasks[task];
        const uint32_t w = t.g / B;
        const Affine<F> *bases = (lev && w) ? lev + (size_t)(w - 1) * n : bases0;      // merged form: window w's points come from level w
        const uint32_t *src = sorted + (size_t)w * n + starts[t.g];
        const uint32_t size = totals[t.g];
        const uint32_t lo = cap + t.seg * SEG;
        const uint32_t hi = lo + SEG < size ? lo + SEG : size;
        Walker<F, MODE> wk;
        for (uint32_t k = lo + threadIdx.x; k < hi; k += 64) {
            const uint32_t e = src[k];
            wk.add(bases[e & 0x7fffffffu], (e >> 31) != 0);
        }
        const Xyzz<F> acc = wk.result();
        static_assert(sizeof(Xyzz<F>) == sizeof(Xyzz<FC>), "layout");
        Xyzz<FC> accc;
        __builtin_memcpy(&accc, &acc, sizeof acc);
        wave_reduce(accc);
        if (threadIdx.x == 0) partials[task] = accc;
    }
}

// one 256-lane workgroup per oversized bucket: fold its partials into buckets[g]
template <class F>
__global__ __launch_bounds__(256) void msm_overflow_fold_kernel(const OverBucket *ob, const MsmDyn *dyn, const Xyzz<F> *partials, Xyzz<F> *buckets) {
    __shared__ Xyzz<F> sh[4];
    const uint32_t n_obs = dyn->n_obs;
    for (uint32_t i = blockIdx.x; i < n_obs; i += gridDim.x) {
        const OverBucket o = ob[i];
        Xyzz<F> acc = Xyzz<F>::inf();
        for (uint32_t k = threadIdx.x; k < o.ntask; k += 256) acc.add(partials[o.task0 + k]);
        block_reduce_256(acc, sh);
        if (threadIdx.x == 0) { Xyzz<F> b = buckets[o.g]; b.add(acc); buckets[o.g] = b; }
        __syncthreads();             // sh is reused by the next bucket
    }
}

// The segment table of the oversized buckets, built by ONE workgroup (there are at most OVER_MAX of them): the list is sorted by
// (bucket, window) -- merged form: the oversized (window, bucket) pairs of one bucket fold into the same sum, one fold
// workgroup for them all --, the segment length is chosen so that all oversized entries give about two waves per SIMD (a lone
// giant bucket -- every scalar equal to 1 meets in one -- becomes a few additions per lane), and every segment gets a task.
__global__ __launch_bounds__(1024) void msm_tasks_kernel(const OverEntry *over, MsmDyn *dyn, uint32_t B, int merged, Task *tasks, uint32_t max_tasks, OverBucket *obs) {
    __shared__ unsigned long long key[OVER_MAX];      // (bucket << 32) | g
    __shared__ uint32_t val[OVER_MAX];                 // size, then segments of the entry
    __shared__ uint32_t t0[OVER_MAX + 1];              // first task of the entry
    __shared__ uint32_t head[OVER_MAX + 1];            // start entry of every fold group
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long xsum[16];
    const uint32_t tid = threadIdx.x;
    uint32_t n = dyn->n_over; if (n > OVER_MAX) n = OVER_MAX;
    if (n == 0) { if (tid == 0) { dyn->seg = SEG_MIN; dyn->n_tasks = 0; dyn->n_obs = 0; } return; }
    const uint32_t cap = dyn->cap;
    uint32_t npow = 1; while (npow < n) npow <<= 1;
    for (uint32_t i = tid; i < npow; i += 1024) {
        if (i < n) { const OverEntry e = over[i]; key[i] = ((unsigned long long)(merged ? e.g % B : e.g) << 32) | e.g; val[i] = e.size; }
        else { key[i] = ~0ull; val[i] = 0; }
    }
    __syncthreads();
    for (uint32_t k = 2; k <= npow; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < npow; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const unsigned long long a = key[i], b = key[l];
                    if ((a > b) == up) { key[i] = b; key[l] = a; const uint32_t v = val[i]; val[i] = val[l]; val[l] = v; }
                }
            }
            __syncthreads();
        }
    // total excess -> segment length
    unsigned long long mine = 0;
    for (uint32_t i = tid; i < n; i += 1024) mine += val[i] - cap;
    mine = wave_sum_u64(mine);
    if ((tid & 63) == 0) xsum[tid >> 6] = mine;
    __syncthreads();
    unsigned long long extra_total = 0;
    for (int w = 0; w < 16; w++) extra_total += xsum[w];
    unsigned long long sg = ((extra_total / 2048 + 63) / 64) * 64;
    const uint32_t SEG = (uint32_t)(sg < SEG_MIN ? SEG_MIN : (sg > SEG_MAX ? SEG_MAX : sg));
    // segments per entry, first task per entry (each lane owns a run of consecutive entries), fold groups
    const uint32_t per = (n + 1023) / 1024, lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0, heads = 0;
    for (uint32_t i = lo; i < hi && lo < n; i++) {
        const uint32_t nt = (val[i] - cap + SEG - 1) / SEG;
        val[i] = nt; sum += nt;
        heads += (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) ? 1u : 0u;
    }
    uint32_t n_tasks, n_obs;
    uint32_t run = block_excl_scan_1024(sum, wsum, &n_tasks);
    uint32_t hrun = block_excl_scan_1024(heads, wsum, &n_obs);
    for (uint32_t i = lo; i < hi && lo < n; i++) {
        t0[i] = run; run += val[i];
        if (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) head[hrun++] = i;
    }
    if (tid == 0) { t0[n] = n_tasks; head[n_obs] = n; }
    __syncthreads();
    if (n_tasks > max_tasks) { if (tid == 0) { dyn->error = 2; dyn->seg = SEG; dyn->n_tasks = 0; dyn->n_obs = 0; } return; }
    for (uint32_t o = tid; o < n_obs; o += 1024) {
        const uint32_t a = head[o], b = head[o + 1];
        obs[o] = OverBucket{(uint32_t)(merged ? key[a] >> 32 : key[a] & 0xffffffffu), t0[a], t0[b] - t0[a]};
    }
    for (uint32_t t = tid; t < n_tasks; t += 1024) {
        uint32_t l = 0, h = n;                 // the entry with t0[l] <= t < t0[l + 1]
        while (h - l > 1) { const uint32_t mid = (l + h) >> 1; if (t0[mid] <= t) l = mid; else h = mid; }
        tasks[t] = Task{(uint32_t)(key[l] & 0xffffffffu), t - t0[l]};
    }
    if (tid == 0) { dyn->seg = SEG; dyn->n_tasks = n_tasks; dyn->n_obs = n_obs; }
}

// ------------------------------------------------------------------------------------------ bucket reduction
// window sum = sum_{slot s} (s+1) * bucket[s].  Lane t owns slots [tL, (t+1)L).
template <class F>
__global__ __launch_bounds__(256) void msm_bucket_reduce_kernel(const Xyzz<F> *buckets, uint32_t B, uint32_t L, uint32_t T,
                                                                uint32_t nblk, Xyzz<F> *winparts) {
    __shared__ Xyzz<F> sh[4];
    const uint32_t w = blockIdx.y;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    Xyzz<F> acc = Xyzz<F>::inf();
    if (t < T) {
        Xyzz<F> run = Xyzz<F>::inf();
        const Xyzz<F> *bk = buckets + (size_t)w * B + (size_t)t * L;
        for (uint32_t s = L; s-- > 0;) {
            run.add(bk[s]);
            acc.add(run);
        }
        // + (t*L) * run
        const uint32_t k = t * L;
        if (k) {
            Xyzz<F> m = Xyzz<F>::inf();
            for (int bit = 31 - __clz(k); bit >= 0; bit--) {
                m = Xyzz<F>::dbl(m);
                if ((k >> bit) & 1) m.add(run);
            }
            acc.add(m);
        }
    }
    block_reduce_256(acc, sh);
    if (threadIdx.x == 0) winparts[(size_t)w * nblk + blockIdx.x] = acc;
}

// The nblk partial sums of a window folded into one point on the device: the host used to add them (128 per multiplication,
// 1.2 ms per proof of host arithmetic AFTER the last kernel -- the GPU idle gap between consecutive proofs).
template <class F>
__global__ __launch_bounds__(256) void msm_fold_partials_kernel(const Xyzz<F> *winparts, uint32_t nblk, Xyzz<F> *out) {
    __shared__ Xyzz<F> sh[4];
    const uint32_t w = blockIdx.x;
    Xyzz<F> acc = Xyzz<F>::inf();
    for (uint32_t b = threadIdx.x; b < nblk; b += 256) acc.add(winparts[(size_t)w * nblk + b]);
    block_reduce_256(acc, sh);
    if (threadIdx.x == 0) out[w] = acc;
}

// ------------------------------------------------------------------------------------------ host driver
#define FK_DBG_ST(ctx, st, name)                                                                  \
    do {                                                                                          \
        if ((ctx)->debug) {                                                                       \
            fprintf(stderr, "[fk] launch %s ...", name); fflush(stderr);                          \
            hipError_t _e = hipStreamSynchronize(st);                                             \
            fprintf(stderr, " %s\n", hipGetErrorString(_e)); fflush(stderr);                      \
        }                                                                                         \
    } while (0)

// every stream of a context, created together and in one order (called by fk_init).  HIP maps streams onto a handful of hardware queues
// in creation order, and streams that share a queue serialise: with lazily created streams the mapping depended on which call a
// process happened to make first -- a process that proved once through fk_prove_r1cs before it pipelined got its copy stream onto the
// hardware queue of the B pair's lane, the upload's completion then sat behind the G2 tail, and the early front of every pipelined
// proof started ~12 ms late (251 against 228 ms per proof on one box; profiles/r05_stream_order_ab.log).  The order here is the one
// the benchmark process of rounds 2-4 happened to create: main (fk_init), copy, auxiliary, the four lanes.
// FK_CU_SPLIT=k (experiment builds; 1 <= k <= 4): k of every 8 compute units are set aside -- the lane streams (sorts, accumulations, tails) are created
// with a CU mask that leaves them out.  Step 1 of "memory-bound work on compute units of its own": what does an accumulation lose when it may
// use only (8 - k) / 8 of the chip?  FK_CU_SPLIT_MODE picks which units of the mask are set aside: 0 = bits with i mod 8 < k, 1 = bits with (i div 8) mod 8 < k
// (how mask bits map to XCDs is not documented for gfx950: with a round-robin mapping mode 0 sets whole XCDs aside and mode 1 units of every XCD, with a
// linear mapping the other way round).
bool cu_masks(fk_ctx *ctx, std::vector<uint32_t> &compute, std::vector<uint32_t> &mem) {
    const int k = tune("FK_CU_SPLIT", 0), mode = tune("FK_CU_SPLIT_MODE", 0);
    if (k < 1 || k > 4) return false;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, ctx->device) != hipSuccess) return false;
    const int ncu = pr.multiProcessorCount, words = (ncu + 31) / 32;
    compute.assign(words, 0); mem.assign(words, 0);
    for (int i = 0; i < ncu; i++) {
        // measured (tools/mulbench/cumask.hip, profiles/r06_cu_mask_semantics.log): mask bit i = XCD i mod 8, unit i div 8 of that XCD; an XCD whose bits are
        // ALL clear is not switched off, it runs unrestricted -- so mode 0 is a no-op; mode 2 sets aside the units j < k of EVERY XCD (j = i div 8)
        const bool m = mode == 0 ? (i % 8) < k : mode == 1 ? ((i / 8) % 8) < k : (i / 8) < k;
        (m ? mem : compute)[i / 32] |= 1u << (i % 32);
    }
    return true;
}

int streams_init(fk_ctx *ctx) {
    std::vector<uint32_t> m_compute, m_mem;
    const bool split = cu_masks(ctx, m_compute, m_mem);
    if (!ctx->copy_st) FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_st, hipStreamNonBlocking));
    if (!ctx->aux) {
        FK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
        FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming));
    }
    for (MsmLane &ln : ctx->lanes) {
        if (ln.st) continue;
        if (split) FK_HIP(ctx, hipExtStreamCreateWithCUMask(&ln.st, (uint32_t)m_compute.size(), ((tune("FK_CU_SPLIT_INVERT", 0) || tune("FK_CU_SPLIT_ALL", 0)) ? m_mem : m_compute).data()));      // (INVERT: sanity check of the mask; ALL: a guest context, everything on the set-aside units)
        else FK_HIP(ctx, hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking));
        FK_HIP(ctx, hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
        FK_HIP(ctx, hipEventCreateWithFlags(&ln.ev_sorted, hipEventDisableTiming));
    }
    if (split && getenv("FK_DEBUG")) fprintf(stderr, "[fk] FK_CU_SPLIT: lane streams masked to %zu words (compute %08x.., set aside %08x..)\n", m_compute.size(), m_compute[0], m_mem[0]);
    return FK_OK;
}

static int lane_init(fk_ctx *ctx, MsmLane &ln) {
    if (ln.st) return FK_OK;
    FK_HIP(ctx, hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking));
    FK_HIP(ctx, hipEventCreateWithFlags(&ln.ev_in, hipEventDisableTiming));
    FK_HIP(ctx, hipEventCreateWithFlags(&ln.ev_sorted, hipEventDisableTiming));
    return FK_OK;
}

static int lane_stage(fk_ctx *ctx, MsmLane &ln, size_t bytes) {     // pinned staging, grow-only; callers own the ordering
    if (bytes <= ln.h_cap) return FK_OK;
    if (ln.h_stage) { FK_HIP(ctx, hipHostFree(ln.h_stage)); ln.h_stage = nullptr; ln.h_cap = 0; }
    FK_HIP(ctx, hipHostMalloc(&ln.h_stage, bytes + (bytes >> 2) + 4096, hipHostMallocDefault));
    ln.h_cap = bytes + (bytes >> 2) + 4096;
    return FK_OK;
}

// Queues one multiplication on a lane: digits, two-pass bucket sort, size ordering, accumulation, oversized buckets, bucket
// reduction and the download of the window sums (with the addition count and the device-side error word behind them) -- all
// without the host waiting anywhere (the only synchronisation left is when a lane's scratch buffers have to GROW, i.e. in
// the first proof of a size).  The five multiplications of a proof are therefore in the lanes' queues microseconds after
// the quotient's kernels, and the GPU decides what runs underneath what.
template <class F>
static int msm_begin(fk_ctx *ctx, const Affine<F> *d_bases, const Fr *d_scalars, size_t n, bool reuse_sort, int *tail_out, hipEvent_t ready, const KeyPre *pre = nullptr) {
    using FC = typename ColdOf<F>::type;   // layout-identical field with an out-of-line multiply
    *tail_out = -1;
    if (n == 0) return FK_OK;
    if (n >= ((size_t)1 << 31)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "msm: n too large");
    int ti = -1;
    for (int i = 0; i < MSM_TAILS; i++) if (!ctx->tails[i].active) { ti = i; break; }
    if (ti < 0) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "msm: too many outstanding multiplications");
    MsmTail &tl = ctx->tails[ti];
    if (!tl.done) FK_HIP(ctx, hipEventCreateWithFlags(&tl.done, hipEventDisableTiming));
    MsmPlan p = make_plan(n, ctx->window_bits);
    bool merged = false;
    if (pre && pre->lev && pre->n == n) {      // the levels were made for exactly this plan; anything else takes the ordinary path
        const MsmPlan pm = make_plan(n, ctx->window_bits, true);
        if (pm.cb == pre->cb && pm.wide == pre->wide && pm.W == pre->W) { p = pm; merged = true; }
    }
    const Affine<F> *d_lev = merged ? (const Affine<F> *)pre->lev : nullptr;
    // lane: the next one in turn, unless this call reuses the previous call's sort (B2 after B1).  With three lanes the
    // multiplication after the G2 one does not queue behind its long overflow / reduction tail (at 2^22 that tail was 5.4 ms
    // during which nothing else ran: 31 % of the proof).
    MsmLane &prev = ctx->lanes[ctx->lane_prev];
    const bool have_sort = reuse_sort && prev.st && prev.last_sort_scalars == (const void *)d_scalars && prev.last_sort_n == n && prev.last_sort_c == p.c && prev.last_merged == merged;
    const int li = have_sort ? ctx->lane_prev : ctx->lane_next;
    MsmLane &ln = ctx->lanes[li];
    FK_TRY(lane_init(ctx, ln));
    // FK_MSM_LANES (experiment builds): fewer lanes -- never while pieces are deferred (sorts-first schedule): its deferred
    // accumulations and tails hold pointers into their lane's buffers, so every multiplication needs a lane of its own there
    static const int t_lanes = tune("FK_MSM_LANES", 0);
    const int n_lanes = (t_lanes >= 1 && t_lanes <= MSM_LANES && !ctx->defer_back && ctx->deferred.empty())
                            ? t_lanes : (ctx->lanes_in_use >= 2 && ctx->lanes_in_use <= MSM_LANES ? ctx->lanes_in_use : MSM_LANES);
    ctx->lane_prev = li; ctx->lane_next = (li + 1) % n_lanes;
    hipStream_t st = ln.st;
    hipStream_t ss = st;                      // the front of the multiplication runs on the lane's stream as well
    if (ready) FK_HIP(ctx, hipStreamWaitEvent(ss, ready, 0));
    else {
        FK_HIP(ctx, hipEventRecord(ln.ev_in, ctx->stream));        // scalars / bases produced on the main stream
        FK_HIP(ctx, hipStreamWaitEvent(ss, ln.ev_in, 0));
    }
    const size_t WB = (size_t)p.W * p.B;
    const uint32_t WR = merged ? 1 : p.W;      // bucket sets to reduce
    // bucket reduction: the flat form (up to 64 buckets per lane, then a double-and-add by the lane's offset).  A hierarchical form
    // (8 buckets per lane, wave-level suffix sums) was built in round 2: a 2.5x shorter serial chain, 1.6x the additions, slower
    // everywhere (171.5 -> 178.3 ms per proof at 2^25) -- removed in round 3.
    const size_t wp_bytes = (size_t)WR * sizeof(Xyzz<F>);                      // ONE point per bucket set (folded on the device)
    const size_t wp_dev_bytes = (size_t)WR * (p.nblk + 1) * sizeof(Xyzz<F>);  // ... behind the nblk partial sums of every set
    // oversized buckets: at most OVER_MAX are tabled; their segment tasks are bounded by max(2048, W n / SEG_MAX) + one per bucket
    const size_t max_tasks = std::max<size_t>(2048, (size_t)p.W * n / SEG_MAX) + OVER_MAX + 64;
    // Growing a buffer frees the old one: everything queued on this lane must be finished first.
    const uint32_t nseg = p.W * p.nhi;
    const size_t max_tiles = (size_t)p.W * ((n + S2_TILE - 1) / S2_TILE) + nseg + 1;
    struct Need { DevBuf *b; size_t bytes; };
    const Need needs[] = {
        {&ln.digits, (size_t)p.W * n * 4}, {&ln.sorted, (size_t)p.W * n * 4}, {&ln.totals, WB * 4}, {&ln.starts, WB * 4},
        {&ln.perm, WB * 4 + SIZE_BINS * 4}, {&ln.overlist, OVER_MAX * sizeof(OverEntry) + sizeof(MsmDyn) + 64}, {have_sort ? &ln.buckets2 : &ln.buckets, WB * sizeof(Xyzz<F>)},
        {&ln.tasktab, max_tasks * sizeof(Task) + OVER_MAX * sizeof(OverBucket) + 64}, {&ln.partials, max_tasks * sizeof(Xyzz<F>)},
        {&ln.s2_cnt1, (size_t)p.W * p.nchunks * p.nhi * 4}, {&ln.s2_seg, ((size_t)nseg * 4 + 2) * 4}, {&ln.s2_cnt2, max_tiles * p.nlo * 4},
        {&ln.s2_tmp_idx, (size_t)p.W * n * 4}, {&ln.s2_tmp_lo, (size_t)p.W * n * 2}};
    bool grow = false;
    for (const Need &nd : needs) grow = grow || nd.bytes > nd.b->cap;
    if (grow) {
        FK_HIP(ctx, hipStreamSynchronize(st));
        for (const Need &nd : needs) FK_HIP(ctx, nd.b->reserve(nd.bytes));
        if (!have_sort) ln.last_sort_scalars = nullptr;
    }
    FK_HIP(ctx, tl.d_wp.reserve(wp_dev_bytes));
    if (wp_bytes + 16 > tl.h_cap) {         // + the additions counter and the error word
        if (tl.h_wp) { FK_HIP(ctx, hipHostFree(tl.h_wp)); tl.h_wp = nullptr; tl.h_cap = 0; }
        FK_HIP(ctx, hipHostMalloc(&tl.h_wp, wp_bytes + (wp_bytes >> 2) + 16, hipHostMallocDefault));
        tl.h_cap = wp_bytes + (wp_bytes >> 2) + 16;
    }
    MsmDyn *dyn = (MsmDyn *)((char *)ln.overlist.p + OVER_MAX * sizeof(OverEntry));       // device-side state of this lane's sort
    unsigned long long *d_adds = &dyn->adds;
    Task *d_tasks = ln.tasktab.as<Task>();
    OverBucket *d_obs = (OverBucket *)((char *)ln.tasktab.p + ((max_tasks * sizeof(Task) + 15) & ~(size_t)15));
    uint32_t *digits = ln.digits.as<uint32_t>(), *sorted = ln.sorted.as<uint32_t>();
    uint32_t *totals = ln.totals.as<uint32_t>(), *starts = ln.starts.as<uint32_t>();
    Xyzz<F> *winparts = tl.d_wp.as<Xyzz<F>>();         // (the bucket buffer is bound in the back half, below)
    uint32_t *perm = ln.perm.as<uint32_t>(), *size_bins = perm + WB;

    if (!have_sort) {
        ln.last_sort_scalars = nullptr;
        hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)(((n + 1) / 2 + 63) / 64)), dim3(64), 0, ss, d_scalars, n, p.cb, p.wide, p.W, digits);
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, ss, "msm_digits");
        FK_HIP(ctx, hipMemsetAsync(dyn, 0, sizeof(MsmDyn), ss));
        uint32_t *cnt1 = ln.s2_cnt1.as<uint32_t>();
        uint32_t *seg_size = ln.s2_seg.as<uint32_t>(), *seg_start = seg_size + nseg, *seg_tiles = seg_start + nseg, *tile_start = seg_tiles + nseg;
        uint32_t *cnt2 = ln.s2_cnt2.as<uint32_t>(), *tmp_idx = ln.s2_tmp_idx.as<uint32_t>();
        uint16_t *tmp_lo = ln.s2_tmp_lo.as<uint16_t>();
        hipLaunchKernelGGL(s2_hist1_kernel, dim3(p.nchunks, p.W), dim3(SORT_THREADS), p.nhi * 4, ss, digits, n, p.chunk, p.nchunks, p.LB, p.nhi, cnt1);
        hipLaunchKernelGGL(s2_prefix1_kernel, dim3(p.W), dim3(1024), p.nhi * 4, ss, cnt1, p.nchunks, p.nhi, seg_size, seg_start, seg_tiles);
        hipLaunchKernelGGL(s2_tile_prefix_kernel, dim3(1), dim3(1024), 0, ss, seg_tiles, nseg, tile_start);
#ifdef FK_S1_NT
        if (tune("FK_S1_THIN", 1)) hipLaunchKernelGGL(s2_scatter1_thin_kernel, dim3(p.nchunks, p.W), dim3(FK_S1_NT), 0, ss, digits, n, p.chunk, p.nchunks, p.LB, p.nhi, cnt1, seg_start, tmp_idx, tmp_lo);
        else
#endif
        hipLaunchKernelGGL(s2_scatter1_n1024_kernel, dim3(p.nchunks, p.W), dim3(1024), 0, ss, digits, n, p.chunk, p.nchunks, p.LB, p.nhi, cnt1, seg_start, tmp_idx, tmp_lo);
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, ss, "msm_sort_pass1");
        // The second pass is launched over the host's BOUND on the tile count (every segment's last tile may be partial:
        // W * ceil(n / tile) + #segments); workgroups beyond the actual count leave at once.  Reading the count back cost a
        // host round trip in the middle of every sort.
        const uint32_t n_tiles = (uint32_t)max_tiles;
        {
            hipLaunchKernelGGL(s2_hist2_kernel, dim3(n_tiles), dim3(256), p.nlo * 4, ss, tmp_lo, n, p.nhi, p.nlo, tile_start, nseg, seg_start, seg_size, cnt2);
        }
        hipLaunchKernelGGL(s2_prefix2_kernel, dim3((nseg + 3) / 4), dim3(256), 0, ss, cnt2, nseg, p.nhi, p.nlo, p.B, tile_start, seg_start, p.cap ? p.cap : 1u, totals, starts, dyn);
        {
            if (p.nlo <= 1024) hipLaunchKernelGGL(s2_scatter2_kernel<1024>, dim3(n_tiles), dim3(1024), 0, ss, tmp_idx, tmp_lo, n, p.nhi, p.nlo, p.B, tile_start, nseg, seg_start, seg_size, cnt2, starts, sorted);
            else if (p.nlo <= 2048) hipLaunchKernelGGL(s2_scatter2_kernel<2048>, dim3(n_tiles), dim3(1024), 0, ss, tmp_idx, tmp_lo, n, p.nhi, p.nlo, p.B, tile_start, nseg, seg_start, seg_size, cnt2, starts, sorted);
            else hipLaunchKernelGGL(s2_scatter2_kernel<4096>, dim3(n_tiles), dim3(1024), 0, ss, tmp_idx, tmp_lo, n, p.nhi, p.nlo, p.B, tile_start, nseg, seg_start, seg_size, cnt2, starts, sorted);
        }
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, ss, "msm_sort_pass2");
        ln.last_sort_scalars = (const void *)d_scalars; ln.last_sort_n = n; ln.last_sort_c = p.c; ln.last_merged = merged;
    }
    // oversized buckets (skewed scalars), size ordering: all on the device (MsmDyn) -- nothing below waits for the host.  With a
    // reused sort (B2 after B1) the lane's state and tables are still valid.
    if (!have_sort) {
        static const int t_many = tune("FK_MSM_OVER_MANY", 2048);
        // "few": every oversized bucket costs a wave per segment plus a 256-lane fold workgroup -- with 4e5 of them, what round 1's
        // WB / 64 allowed at 2^25, the overflow + fold kernels took 70 ms of a proof whose witness held each value 341 times
        const uint32_t many = (uint32_t)std::min<size_t>(std::max<size_t>((size_t)t_many, WB / 8192), OVER_MAX - 64);
        hipLaunchKernelGGL(msm_cap_kernel, dim3(1), dim3(64), 0, ss, dyn, p.cap ? p.cap : 1u, many);
        hipLaunchKernelGGL(msm_over_list_kernel, dim3((unsigned)((WB + 255) / 256)), dim3(256), 0, ss, totals, WB, ln.overlist.as<OverEntry>(), dyn);
        hipLaunchKernelGGL(msm_tasks_kernel, dim3(1), dim3(1024), 0, ss, ln.overlist.as<OverEntry>(), dyn, p.B, merged ? 1 : 0, d_tasks, (uint32_t)max_tasks, d_obs);
        // size-ordered bucket -> lane assignment
        FK_HIP(ctx, hipMemsetAsync(size_bins, 0, SIZE_BINS * 4, ss));
        if (merged) {     // one bucket set: order its B buckets by their length over all windows (perm[0, B); lengths kept behind it)
            uint32_t *mt = perm + p.B;
            hipLaunchKernelGGL(msm_merge_totals_kernel, dim3(std::min<uint32_t>((p.B + 255) / 256, 1024)), dim3(256), 0, ss, totals, p.B, p.W, dyn, mt, d_adds);
            hipLaunchKernelGGL(msm_size_hist_kernel, dim3((unsigned)std::min<size_t>((p.B + 255) / 256, 1024)), dim3(256), 0, ss, mt, (size_t)p.B, dyn, p.W, size_bins, (unsigned long long *)nullptr);
            hipLaunchKernelGGL(msm_size_scan_kernel, dim3(1), dim3(SIZE_BINS), 0, ss, size_bins);
            hipLaunchKernelGGL(msm_size_scatter_kernel, dim3((unsigned)((p.B + 1023) / 1024)), dim3(1024), 0, ss, mt, (size_t)p.B, dyn, p.W, size_bins, perm);
        } else {
            hipLaunchKernelGGL(msm_size_hist_kernel, dim3((unsigned)std::min<size_t>((WB + 255) / 256, 1024)), dim3(256), 0, ss, totals, WB, dyn, 1u, size_bins, d_adds);
            hipLaunchKernelGGL(msm_size_scan_kernel, dim3(1), dim3(SIZE_BINS), 0, ss, size_bins);
            hipLaunchKernelGGL(msm_size_scatter_kernel, dim3((unsigned)((WB + 1023) / 1024)), dim3(1024), 0, ss, totals, WB, dyn, 1u, size_bins, perm);
        }
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, ss, "msm_size_order");
        if (ctx->debug) {
            MsmDyn h{};
            FK_HIP(ctx, hipMemcpy(&h, dyn, sizeof h, hipMemcpyDeviceToHost));
            fprintf(stderr, "[fk] msm n=%zu c=%u (W=%u: %u x %u bits + %u x %u bits) cap=%u (plan %u): %u oversized buckets, %u tasks of %u entries, %u fold groups\n", n, p.c,
                    p.W, p.wide, p.cb + 1, p.W - p.wide, p.cb, h.cap, p.cap, h.n_over, h.n_tasks, h.seg, h.n_obs);
            fflush(stderr);
        }
    }

    // ---- from here on nothing waits for the host
    if (!have_sort) { FK_HIP(ctx, hipEventRecord(ln.ev_sorted, ss)); ln.ev_sorted_valid = true; }
    tl.active = true; tl.cb = p.cb; tl.wide = p.wide; tl.W = WR; tl.nblk = 1;     // merged: one "window" of weight 1
    *tail_out = ti;
    // The back of the multiplication in two pieces -- the accumulation, and the tail (oversized buckets, reduction, download) --
    // queued now, or by msm_run_deferred (ctx->defer_back): all accumulations first, then all tails, so that on the B pair's lane
    // the G2 accumulation follows the G1 one at once and both tails come behind (B2 has a bucket buffer of its own for that).
    MsmLane *lnp = &ln; MsmTail *tlp = &tl;
    // FK_MSM_LAZY (default 1): accumulators in the lazily reduced form [0, 2p) (Walker<F, 2>); 0 = canonical after every product
    static const int t_lazy = tune("FK_MSM_LAZY", 1);
    constexpr bool IS_G1 = std::is_same<F, Fq>::value;
    const bool lazy = t_lazy != 0;
    constexpr int MINW_ = IS_G1 ? 4 : 2;
    // bound late (inside the pieces): a multiplication begun on this lane in between may have GROWN a buffer, i.e. moved it
    auto bucket_buf = [=]() -> Xyzz<F> * { return (have_sort ? lnp->buckets2 : lnp->buckets).template as<Xyzz<F>>(); };
    auto back_acc = [=]() -> int {
        Xyzz<F> *buckets = bucket_buf();
        std::vector<EventPair> &evv = (sizeof(F) == sizeof(Fq)) ? ctx->ev_acc : ctx->ev_acc2;
        FK_TRY(stats_begin(ctx, evv, (uint64_t)n, st));
        if (merged) {
            if (lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_accumulate_merged_kernel<F, MINW_, 2>), dim3((p.B + 255) / 256), dim3(256), 0, st, d_bases, d_lev, sorted, n,
                                              starts, totals, p.B, p.W, dyn, perm, perm + p.B, buckets);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_accumulate_merged_kernel<F, MINW_, 0>), dim3((p.B + 255) / 256), dim3(256), 0, st, d_bases, d_lev, sorted, n,
                                    starts, totals, p.B, p.W, dyn, perm, perm + p.B, buckets);
        } else {
            if (lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_accumulate_kernel<F, MINW_, 2>), dim3((unsigned)((WB + 255) / 256)), dim3(256), 0, st, d_bases, sorted, n,
                                              starts, totals, p.B, p.W, dyn, perm, buckets);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_accumulate_kernel<F, MINW_, 0>), dim3((unsigned)((WB + 255) / 256)), dim3(256), 0, st, d_bases, sorted, n,
                                    starts, totals, p.B, p.W, dyn, perm, buckets);
        }
        FK_HIP(ctx, hipGetLastError());
        FK_TRY(stats_end(ctx, evv, st));
        if (!ctx->ev_acc_done) FK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_acc_done, hipEventDisableTiming));
        FK_HIP(ctx, hipEventRecord(ctx->ev_acc_done, st)); ctx->ev_acc_done_valid = true;
        FK_DBG_ST(ctx, st, "msm_accumulate");
        return FK_OK;
    };
    auto back_tail = [=]() -> int {
        MsmLane &ln = *lnp; MsmTail &tl = *tlp;
        Xyzz<F> *buckets = bucket_buf();
        // oversized buckets: fixed grids looping over the device-built tables (they leave at once when there is nothing to do)
        if (lazy) hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_overflow_kernel<F, FC, 2>), dim3(2048), dim3(64), 0, st,
                                          d_bases, d_lev, sorted, n, starts, totals, p.B, dyn, d_tasks, ln.partials.as<Xyzz<FC>>());
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_overflow_kernel<F, FC, 0>), dim3(2048), dim3(64), 0, st,
                                d_bases, d_lev, sorted, n, starts, totals, p.B, dyn, d_tasks, ln.partials.as<Xyzz<FC>>());
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, st, "msm_overflow");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_overflow_fold_kernel<F>), dim3(256), dim3(256), 0, st, d_obs, dyn, ln.partials.as<Xyzz<F>>(), buckets);
        FK_HIP(ctx, hipGetLastError());
        FK_DBG_ST(ctx, st, "msm_overflow_fold");
        const Xyzz<F> *wp_src = winparts;
        {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_bucket_reduce_kernel<F>), dim3(p.nblk, WR), dim3(256), 0, st,
                               buckets, p.B, p.L, p.T, p.nblk, winparts);
            Xyzz<F> *folded = winparts + (size_t)WR * p.nblk;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_fold_partials_kernel<F>), dim3(WR), dim3(256), 0, st, winparts, p.nblk, folded);
            wp_src = folded;
        }
        FK_HIP(ctx, hipGetLastError());
        FK_HIP(ctx, hipMemcpyAsync(tl.h_wp, wp_src, wp_bytes, hipMemcpyDeviceToHost, st));
        FK_HIP(ctx, hipMemcpyAsync((char *)tl.h_wp + wp_bytes, d_adds, 8, hipMemcpyDeviceToHost, st));
        FK_HIP(ctx, hipMemcpyAsync((char *)tl.h_wp + wp_bytes + 8, &dyn->error, 4, hipMemcpyDeviceToHost, st));
        FK_HIP(ctx, hipEventRecord(tl.done, st));
        FK_DBG_ST(ctx, st, "msm_bucket_reduce");
        return FK_OK;
    };
    if (ctx->defer_back) {
        ctx->deferred.push_back(back_acc);
        ctx->deferred_tails.push_back(back_tail);
        return FK_OK;
    }
    FK_TRY(back_acc());
    return back_tail();
}

// queues the deferred pieces in the order their multiplications were begun -- all accumulations, then all tails (the lane
// streams first wait for `after`, if given)
int msm_run_deferred(fk_ctx *ctx, hipEvent_t after) {
    if (after) for (MsmLane &ln : ctx->lanes) if (ln.st) FK_HIP(ctx, hipStreamWaitEvent(ln.st, after, 0));
    std::vector<std::function<int()>> accs, tails;
    accs.swap(ctx->deferred); tails.swap(ctx->deferred_tails);
    for (auto &f : accs) FK_TRY(f());
    for (auto &f : tails) FK_TRY(f());
    return FK_OK;
}

template <class F>
static int msm_end(fk_ctx *ctx, int tail, Xyzz<F> *out) {
    *out = Xyzz<F>::inf();
    if (tail < 0) return FK_OK;
    if (tail >= MSM_TAILS || !ctx->tails[tail].active) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "msm: bad tail handle");
    MsmTail &tl = ctx->tails[tail];
    tl.active = false;
    FK_HIP(ctx, hipEventSynchronize(tl.done));
    const Xyzz<F> *wp = (const Xyzz<F> *)tl.h_wp;
    { const char *extra = (const char *)tl.h_wp + (size_t)tl.W * tl.nblk * sizeof(Xyzz<F>);
      uint64_t adds; uint32_t err;
      memcpy(&adds, extra, 8); memcpy(&err, extra + 8, 4);
      ctx->acc_adds[sizeof(F) == sizeof(Fq) ? 0 : 1] += adds;
      if (err) FK_SET_ERR(ctx, FK_ERR_HIP, "msm: the oversized-bucket tables overflowed (%s)", err == 1 ? "more than OVER_MAX buckets above the cap" : "more segment tasks than the bound"); }
    // Horner over windows, most significant first
    Xyzz<F> acc = Xyzz<F>::inf();
    for (uint32_t w = tl.W; w-- > 0;) {
        const uint32_t cw = tl.cb + (w < tl.wide ? 1 : 0);       // acc holds the windows above w, relative to w's top bit
        for (uint32_t k = 0; k < cw; k++) acc = Xyzz<F>::dbl(acc);
        for (uint32_t b = 0; b < tl.nblk; b++) acc.add(wp[(size_t)w * tl.nblk + b]);
    }
    *out = acc;
    if (ctx->debug) { fprintf(stderr, "[fk] msm host horner done\n"); fflush(stderr); }
    return FK_OK;
}

int msm_sync(fk_ctx *ctx) {
    if (ctx->aux) FK_HIP(ctx, hipStreamSynchronize(ctx->aux));
    for (MsmLane &ln : ctx->lanes) if (ln.st) FK_HIP(ctx, hipStreamSynchronize(ln.st));
    return FK_OK;
}

void msm_abandon(fk_ctx *ctx) {
    ctx->wit_active = false;
    ctx->early.done = false;
    ctx->defer_back = false; ctx->deferred.clear(); ctx->deferred_tails.clear();
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    for (int i = 0; i < MSM_TAILS; i++) ctx->tails[i].active = false;
    for (MsmLane &ln : ctx->lanes) { if (ln.st) (void)hipStreamSynchronize(ln.st); ln.last_sort_scalars = nullptr; }
}

void msm_release(fk_ctx *ctx) {
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); ctx->aux = nullptr; }
    if (ctx->ev_aux) { (void)hipEventDestroy(ctx->ev_aux); ctx->ev_aux = nullptr; }
    if (ctx->ev_main) { (void)hipEventDestroy(ctx->ev_main); ctx->ev_main = nullptr; }
    if (ctx->ev_z) { (void)hipEventDestroy(ctx->ev_z); ctx->ev_z = nullptr; }
    if (ctx->ev_upload_gate) { (void)hipEventDestroy(ctx->ev_upload_gate); ctx->ev_upload_gate = nullptr; }
    for (hipEvent_t &e : ctx->ev_chunk) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ctx->ev_acc_done) { (void)hipEventDestroy(ctx->ev_acc_done); ctx->ev_acc_done = nullptr; ctx->ev_acc_done_valid = false; }
    for (MsmLane &ln : ctx->lanes) {
        if (ln.st) (void)hipStreamSynchronize(ln.st);
        for (DevBuf *b : {&ln.digits, &ln.sorted, &ln.totals, &ln.starts, &ln.perm, &ln.overlist, &ln.tasktab, &ln.partials, &ln.s2_cnt1, &ln.s2_seg,
                          &ln.s2_cnt2, &ln.s2_tmp_idx, &ln.s2_tmp_lo, &ln.buckets, &ln.buckets2})
            b->release();
        if (ln.h_stage) (void)hipHostFree(ln.h_stage);
        if (ln.ev_in) (void)hipEventDestroy(ln.ev_in);
        if (ln.ev_sorted) (void)hipEventDestroy(ln.ev_sorted);
        if (ln.st) (void)hipStreamDestroy(ln.st);
        ln = MsmLane();
    }
    for (int i = 0; i < MSM_TAILS; i++) {
        MsmTail &tl = ctx->tails[i];
        if (tl.done) (void)hipEventDestroy(tl.done);
        if (tl.h_wp) (void)hipHostFree(tl.h_wp);
        tl.d_wp.release();
        tl = MsmTail();
    }
}

int msm_g1_begin(fk_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, int *tail, hipEvent_t ready, const KeyPre *pre) {
    return msm_begin<Fq>(ctx, d_bases, d_scalars, n, false, tail, ready, pre);
}
int msm_g1_end(fk_ctx *ctx, int tail, G1Xyzz *out) { return msm_end<Fq>(ctx, tail, out); }
int msm_g2_begin(fk_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, bool reuse_sort, int *tail, hipEvent_t ready, const KeyPre *pre) {
    return msm_begin<Fq2>(ctx, d_bases, d_scalars, n, reuse_sort, tail, ready, pre);
}
int msm_g2_end(fk_ctx *ctx, int tail, G2Xyzz *out) { return msm_end<Fq2>(ctx, tail, out); }

template <class F>
static int msm_run(fk_ctx *ctx, const Affine<F> *d_bases, const Fr *d_scalars, size_t n, Xyzz<F> *out, bool reuse_sort = false, const KeyPre *pre = nullptr) {
    int tail = -1;
    *out = Xyzz<F>::inf();
    FK_TRY(msm_begin<F>(ctx, d_bases, d_scalars, n, reuse_sort, &tail, nullptr, pre));
    return msm_end<F>(ctx, tail, out);
}

int msm_g1_dev(fk_ctx *ctx, const G1Affine *d_bases, const Fr *d_scalars, size_t n, G1Xyzz *out, const KeyPre *pre) {
    return msm_run<Fq>(ctx, d_bases, d_scalars, n, out, false, pre);
}
int msm_g2_dev(fk_ctx *ctx, const G2Affine *d_bases, const Fr *d_scalars, size_t n, G2Xyzz *out, bool reuse_sort, const KeyPre *pre) {
    return msm_run<Fq2>(ctx, d_bases, d_scalars, n, out, reuse_sort, pre);
}

// ------------------------------------------------------------------------------------------ fixed-base precomputation of a key
// name: the key array (for the message); require: FK_MSM_PRECOMP=require -- fail instead of falling back when HBM is short
template <class F>
static int precompute_levels(fk_ctx *ctx, const Affine<F> *d_bases, size_t n, KeyPre *out, const char *name, bool require, size_t reserve) {
    *out = KeyPre();
    // Arrays below 2^21 points (rounded as the window rule rounds) keep the ordinary path.  Rounds 1-3 drew the line at 2^24 (measured
    // per proof with / without levels -- 2^20: 15.5 / 13.1 ms, 2^22: 29.9 / 25.8, 2^23: 45.7 / 44.1, 2^24: 79.0 / 80.6, 2^25: 139.5 /
    // 148.2): the merged form lost below it because its ONE bucket set was reduced by 64 buckets per lane -- 32 workgroups for 2^19
    // buckets, twice the time of the W-set form's reduction (profiles/r04_shard_levels_kernel_stats.log).  With the short chain of
    // make_plan (L = 8 / 16 for small bucket sets) the levels pay from ~2^21 points on (round 4, host-witness ms per proof without /
    // with: 2^22 synthetic 21.3 / 20.6, 2^23 37.0 / 36.3, 256 transactions 50.9 / 48.7, 512 transactions 95.5 / 89.6; 2^20: 9.8 / 10.2 --
    // still a loss), which also keeps them on the 1/4 and 1/8 shards of a 2^25 key (section 4.4).  (Read per call: the tests lower it.)
    const char *e = getenv("FK_MSM_PRE_MIN_LOG2");
    const int min_lg = e ? atoi(e) : 21;
    if (min_lg < 6 || min_lg > 40 || n + n / 2 < ((size_t)1 << min_lg)) return FK_OK;
    const MsmPlan p = make_plan(n, ctx->window_bits, true);
    if (p.W < 2) return FK_OK;
    const size_t bytes = (size_t)(p.W - 1) * n * sizeof(Affine<F>);
    size_t fr = 0, tot = 0;
    FK_HIP(ctx, hipMemGetInfo(&fr, &tot));
    // leave room for what a proof with this key will allocate later (`reserve`, computed once per key: key_precompute)
    void *lev = nullptr;
    const bool fits = bytes + reserve <= fr;
    if (!fits || hipMalloc(&lev, bytes) != hipSuccess) {
        (void)hipGetLastError();
        // Not an error by default -- the array keeps the ordinary W-bucket-set path (same bytes, ~7-10 % slower at 2^25) -- but never
        // silent: the text stays in fk_last_error (a second key or another process on the GPU is the usual cause).
        char msg[320];
        snprintf(msg, sizeof msg, "key: fixed-base levels of %s (%zu points, %.1f GiB) do not fit into the %.1f GiB of free HBM", name, n,
                 (double)bytes / (double)(1ull << 30), (double)fr / (double)(1ull << 30));
        if (require) { ctx->err = std::string(msg) + " and FK_MSM_PRECOMP=require"; return FK_ERR_OOM; }
        ctx->err += (ctx->err.empty() ? "warning: " : "; ") + std::string(msg) + " -- that array takes the slower W-bucket-set path";
        return FK_OK;
    }
    const Affine<F> *prev = d_bases;
    for (uint32_t w = 1; w < p.W; w++) {
        Affine<F> *cur = (Affine<F> *)lev + (size_t)(w - 1) * n;
        const uint32_t bits = p.cb + ((w - 1) < p.wide ? 1 : 0);       // width of window w-1
        constexpr int NB = sizeof(F) == sizeof(Fq) ? 4 : 2;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(msm_level_kernel<F, NB>), dim3((unsigned)(((n + NB - 1) / NB + 255) / 256)), dim3(256), 0, ctx->stream, prev, n, bits, cur);
        prev = cur;
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipFree(lev); FK_SET_ERR(ctx, FK_ERR_HIP, "msm: level precomputation failed"); }
    out->lev = lev; out->n = n; out->cb = p.cb; out->wide = p.wide; out->W = p.W;
    return FK_OK;
}

void key_pre_free(fk_key *k) {
    for (KeyPre *p : {&k->pre_h, &k->pre_l, &k->pre_a, &k->pre_b1, &k->pre_b2}) { if (p->lev) (void)hipFree(p->lev); *p = KeyPre(); }
}

// FK_MSM_PRECOMP: unset / 1 = derive the levels that fit (a skipped array leaves a warning in fk_last_error), 0 = none,
// require = fail with FK_ERR_OOM instead of degrading.
// What proofs with key k will still allocate in this context (see key_precompute), times the ranks that share the device.
static size_t proof_scratch_still_needed(fk_ctx *ctx, const fk_key *k) {
    // What proofs with this key allocate AFTER the key is loaded, and the levels must leave free (round 4: the rule used to be 640 B x
    // 4 lanes x the points of the array in hand -- 172 GB for the h array of a 2^26 domain, so the reference-size system of bench.py
    // lost h's levels although everything fits):
    //   * lane scratch, one lane per multiplication (msm_begin_t: digits, sorted, two staging arrays = 14 B per digit, W <= 13 digits
    //     per scalar) + per lane the bucket sets, counters and task tables (3.2 GB G1 / 6.4 GB G2 at c = 22, from the plan of the
    //     largest array);
    //   * the quotient's vectors (a, b, c, h, staging, scalar vectors: 10 of m / ranks x 32 B) and tables (two full twiddle tables, four scale
    //     tables: 6 of m x 32 B on every rank);
    //   * two witness slots and 8 GB for the caller (a resident constraint system, the bench's own buffers).
    const size_t pts = (size_t)(k->h_hi - k->h_lo) + (k->l_hi - k->l_lo) + (k->a_hi - k->a_lo) + std::max(k->b_hi - k->b_lo, k->b2_hi - k->b2_lo);
    const size_t nv = (size_t)k->num_input + k->num_aux;
    const size_t nmax = std::max<size_t>(std::max<size_t>(k->h_hi - k->h_lo, k->l_hi - k->l_lo), std::max<size_t>(k->a_hi - k->a_lo, std::max(k->b_hi - k->b_lo, k->b2_hi - k->b2_lo)));
    const MsmPlan pl = make_plan(nmax, ctx->window_bits, true);
    const size_t lane_fixed = (size_t)pl.W * pl.B * (MSM_LANES * (sizeof(Xyzz<Fq>) + 16) + sizeof(Xyzz<Fq2>)) + ((size_t)1 << 30);   // bucket sets (G1 per lane, G2 once), counters, tables
    // The quotient's share of this shard follows from the h slice it holds: all of h = the whole quotient runs here (a lone
    // shard, or rank 0 of the "quotient on rank 0" split: 10 vectors + 6 tables of m elements); no h at all = no quotient buffers (the
    // other ranks of that split); a block of the domain = the cut transforms (vectors of m / ranks elements, tables of m on every rank).
    const uint64_t h_held = k->h_hi - k->h_lo;
    const size_t quot = h_held == 0 ? 0 : h_held >= k->n_h ? (size_t)k->m * 32 * (6 + 10) : (size_t)k->m * 32 * 6 + (size_t)k->m / k->shard_count * 32 * 10;
    const size_t need = pts * 14 * 13 + lane_fixed + quot + 2 * nv * 32;
    // ... of which this context may hold a part already (grow-only buffers of earlier, possibly larger proofs: they are reused)
    size_t have = 0;
    for (const MsmLane &ln : ctx->lanes)
        for (const DevBuf *b : {&ln.digits, &ln.sorted, &ln.totals, &ln.starts, &ln.perm, &ln.overlist, &ln.tasktab, &ln.partials, &ln.s2_cnt1, &ln.s2_seg,
                                &ln.s2_cnt2, &ln.s2_tmp_idx, &ln.s2_tmp_lo, &ln.buckets, &ln.buckets2}) have += b->cap;
    for (const DevBuf *b : {&ctx->ntt_s1, &ctx->ntt_s2, &ctx->ntt_io, &ctx->hbuf, &ctx->sc_a, &ctx->sc_b, &ctx->stage_a, &ctx->stage_b, &ctx->stage_c, &ctx->stage_z,
                            &ctx->stage_d, &ctx->wslot[0].buf, &ctx->wslot[1].buf}) have += b->cap;
    // ranks of one fk_multi that share a GPU (fk_init_devices with a device named several times) each need this much again
    return (need > have ? need - have : 0) * (size_t)std::max(1, ctx->co_tenants);
}

// HBM free now minus what proofs with this key still allocate (and 2 GB): negative = the levels were planned before something else was
// placed in HBM and no longer leave room -- fk_key_derive_levels plans them again against what is free now.
int key_levels_headroom(fk_ctx *ctx, const fk_key *k, int64_t *bytes) {
    size_t fr = 0, tot = 0;
    FK_HIP(ctx, hipMemGetInfo(&fr, &tot));
    *bytes = (int64_t)fr - (int64_t)proof_scratch_still_needed(ctx, k) - ((int64_t)2 << 30);
    return FK_OK;
}

int key_precompute(fk_ctx *ctx, fk_key *k) {
    const char *e = getenv("FK_MSM_PRECOMP");
    const bool require = e && !strcmp(e, "require");
    const int on = e && !require ? atoi(e) : 1;
    key_pre_free(k);
    ctx->err.clear();
    if (!on) return FK_OK;
    const size_t reserve = proof_scratch_still_needed(ctx, k) + ((size_t)8 << 30);       // + 8 GB for the caller (a resident constraint system, the bench's own buffers)
    // WHICH arrays get levels when they do not all fit (2^26: everything but one; 2^27: one or two): the set with the most accumulation
    // work on the merged path that fits -- a knapsack over at most five items, solved by enumeration.  An array's worth is its points
    // times the cost of one of its additions (a G2 addition = FK_G2_WORK G1 additions), its price (W - 1) levels of 64 / 128 bytes per
    // point: G2 buys 1.4 x the work per byte of G1.  b_g1 and b_g2 over the same scalars share one sort, hence one plan: both or
    // neither.  (Rounds 1-4 took the arrays in a fixed order h, l, b, a while they fitted: at 2^26 that dropped a, the cheapest, and at
    // 2^27 kept only a -- VERDICT r4 #8.)
    struct Cand { const char *name; int which; size_t n, bytes; double worth; };       // which: 0 h, 1 l, 2 a, 3 b_g1, 4 b_g2, 5 the b pair
    std::vector<Cand> cand;
    const bool b_shared = k->b2_lo == k->b_lo && k->b2_hi == k->b_hi && k->d_b1 && k->d_b2;          // B1 and B2 over the same scalars share one sort: same plan or none
    auto level_bytes = [&](size_t n, size_t pt) -> size_t {
        const char *e2 = getenv("FK_MSM_PRE_MIN_LOG2");
        const int min_lg = e2 ? atoi(e2) : 21;
        if (min_lg < 6 || min_lg > 40 || n + n / 2 < ((size_t)1 << min_lg)) return 0;
        const MsmPlan p = make_plan(n, ctx->window_bits, true);
        return p.W < 2 ? 0 : (size_t)(p.W - 1) * n * pt;
    };
    auto add_cand = [&](const char *name, int which, const void *d, size_t n, size_t pt, double w) {
        const size_t b = d ? level_bytes(n, pt) : 0;
        if (b) cand.push_back({name, which, n, b, w * (double)n});
    };
    add_cand("h", 0, k->d_h, k->h_hi - k->h_lo, 64, 1.0);
    add_cand("l", 1, k->d_l, k->l_hi - k->l_lo, 64, 1.0);
    add_cand("a", 2, k->d_a, k->a_hi - k->a_lo, 64, 1.0);
    if (b_shared) {
        const size_t nb = k->b_hi - k->b_lo, bb = level_bytes(nb, 64 + 128);
        if (bb) cand.push_back({"b_g1 + b_g2", 5, nb, bb, (1.0 + (double)FK_G2_WORK) * (double)nb});
    } else {
        add_cand("b_g1", 3, k->d_b1, k->b_hi - k->b_lo, 64, 1.0);
        add_cand("b_g2", 4, k->d_b2, k->b2_hi - k->b2_lo, 128, (double)FK_G2_WORK);
    }
    size_t fr = 0, tot = 0;
    FK_HIP(ctx, hipMemGetInfo(&fr, &tot));
    const size_t budget = fr > reserve ? fr - reserve : 0;
    uint32_t best = 0; double best_w = -1; size_t best_b = 0;
    for (uint32_t mask = 0; mask < (1u << cand.size()); mask++) {
        size_t b = 0; double w = 0;
        for (size_t i = 0; i < cand.size(); i++) if ((mask >> i) & 1) { b += cand[i].bytes; w += cand[i].worth; }
        if (b <= budget && (w > best_w || (w == best_w && b < best_b))) { best = mask; best_w = w; best_b = b; }
    }
    for (size_t i = 0; i < cand.size(); i++) {
        const Cand &c = cand[i];
        if (!((best >> i) & 1)) {
            // Not an error by default -- the array keeps the ordinary W-bucket-set path (same bytes, ~7-10 % slower at 2^25) -- but never
            // silent: the text stays in fk_last_error (a second key or another process on the GPU is the usual cause).
            char msg[320];
            snprintf(msg, sizeof msg, "key: fixed-base levels of %s (%zu points, %.1f GiB) left out: %.1f GiB of HBM are free for levels and the arrays chosen instead carry more work",
                     c.name, c.n, (double)c.bytes / (double)(1ull << 30), (double)budget / (double)(1ull << 30));
            if (require) { ctx->err = std::string(msg) + " and FK_MSM_PRECOMP=require"; return FK_ERR_OOM; }
            ctx->err += (ctx->err.empty() ? "warning: " : "; ") + std::string(msg) + " -- that array takes the slower W-bucket-set path";
            continue;
        }
        switch (c.which) {
            case 0: FK_TRY(precompute_levels<Fq>(ctx, k->d_h, c.n, &k->pre_h, "h", require, 0)); break;
            case 1: FK_TRY(precompute_levels<Fq>(ctx, k->d_l, c.n, &k->pre_l, "l", require, 0)); break;
            case 2: FK_TRY(precompute_levels<Fq>(ctx, k->d_a, c.n, &k->pre_a, "a", require, 0)); break;
            case 3: FK_TRY(precompute_levels<Fq>(ctx, k->d_b1, c.n, &k->pre_b1, "b_g1", require, 0)); break;
            case 4: FK_TRY(precompute_levels<Fq2>(ctx, k->d_b2, c.n, &k->pre_b2, "b_g2", require, 0)); break;
            default:
                FK_TRY(precompute_levels<Fq2>(ctx, k->d_b2, c.n, &k->pre_b2, "b_g2", require, 0));
                FK_TRY(precompute_levels<Fq>(ctx, k->d_b1, c.n, &k->pre_b1, "b_g1", require, 0));
                if (k->pre_b2.lev && !k->pre_b1.lev) { (void)hipFree(k->pre_b2.lev); k->pre_b2 = KeyPre(); }       // (an allocation failed after all: neither keeps them)
                if (k->pre_b1.lev && !k->pre_b2.lev) { (void)hipFree(k->pre_b1.lev); k->pre_b1 = KeyPre(); }
        }
    }
    return FK_OK;
}

// ------------------------------------------------------------------------------------------ generators (bench/test inputs)
static __host__ __device__ inline uint64_t splitmix64(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Every lane derives a 254-bit multiplier k_t, computes k_t * G by double-and-add and then walks
// P, P + D, P + 2D, ... (D = step point) writing PER affine points -- valid, distinct-looking points.
template <class F>
__global__ __launch_bounds__(64) void gen_points_kernel(Affine<F> gen, Affine<F> step, Affine<F> *out, size_t n, uint32_t per, uint64_t seed) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t lo = t * per;
    if (lo >= n) return;
    uint64_t s = seed ^ (0xA0761D6478BD642Full * (t + 1));
    uint32_t k[8];
    for (int i = 0; i < 4; i++) { uint64_t v = splitmix64(s); k[2 * i] = (uint32_t)v; k[2 * i + 1] = (uint32_t)(v >> 32); }
    k[7] &= 0x0fffffffu;  // < 2^252 < r
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int i = 251; i >= 0; i--) {
        acc = Xyzz<F>::dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) acc.add_mixed(gen);
    }
    size_t hi = lo + per < n ? lo + per : n;
    for (size_t i = lo; i < hi; i++) {
        out[i] = acc.to_affine();
        acc.add_mixed(step);
    }
}

static G1Affine g1_generator() { G1Affine g; g.x = Fq::from_u64(1); g.y = Fq::from_u64(2); return g; }
static Fq fq_words(const uint32_t (&w)[8]) { Fq t; for (int i = 0; i < 8; i++) t.v[i] = w[i]; return t; }
static G2Affine g2_generator() {
    const uint32_t x0[8] = FK_G2_GEN_X0, x1[8] = FK_G2_GEN_X1, y0[8] = FK_G2_GEN_Y0, y1[8] = FK_G2_GEN_Y1;
    G2Affine g;
    g.x.c0 = fq_words(x0); g.x.c1 = fq_words(x1);
    g.y.c0 = fq_words(y0); g.y.c1 = fq_words(y1);
    return g;
}

template <class F>
static int gen_points(fk_ctx *ctx, const Affine<F> &gen, Affine<F> *d_out, size_t n, uint64_t seed) {
    if (!n) return FK_OK;
    // step point D = kd * G on the host
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t kd[8];
    for (int i = 0; i < 4; i++) { uint64_t v = splitmix64(s); kd[2 * i] = (uint32_t)v; kd[2 * i + 1] = (uint32_t)(v >> 32); }
    kd[7] &= 0x0fffffffu; kd[0] |= 1;
    Affine<F> step = Xyzz<F>::mul_scalar(Xyzz<F>::from_affine(gen), kd).to_affine();
    const uint32_t per = 16;
    const size_t threads = (n + per - 1) / per;
    using FC = typename ColdOf<F>::type;
    Affine<FC> genc, stepc;
    static_assert(sizeof(genc) == sizeof(gen), "layout");
    memcpy(&genc, &gen, sizeof gen); memcpy(&stepc, &step, sizeof step);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gen_points_kernel<FC>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, ctx->stream, genc, stepc,
                       reinterpret_cast<Affine<FC> *>(d_out), n, per, seed);
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}
int gen_points_g1(fk_ctx *ctx, G1Affine *d_out, size_t n, uint64_t seed) { return gen_points<Fq>(ctx, g1_generator(), d_out, n, seed); }
int gen_points_g2(fk_ctx *ctx, G2Affine *d_out, size_t n, uint64_t seed) { return gen_points<Fq2>(ctx, g2_generator(), d_out, n, seed); }

__global__ void gen_scalars_kernel(Fr *out, size_t n, uint64_t seed, int kind) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed ^ (0xE7037ED1A0B428DBull * (i + 1));
    Fr v;
    for (int k = 0; k < 4; k++) { uint64_t x = splitmix64(s); v.v[2 * k] = (uint32_t)x; v.v[2 * k + 1] = (uint32_t)(x >> 32); }
    v.v[7] &= 0x3fffffffu;                 // 254 random bits, then one conditional subtraction: (almost) uniform mod r,
    v = Fr::reduce_once(v);                // so that the top window sees the load real scalars give it
    if (kind == 1) {
        uint64_t sel = splitmix64(s);
        if (sel & 1) { for (int k = 0; k < 8; k++) v.v[k] = 0; v.v[0] = (uint32_t)((sel >> 1) & 1); }
    } else if (kind == 2) {                // the benchmark witness's mix: 5.3 % zeros, 2.5 % ones, the rest dense and pairwise DISTINCT
        const uint32_t sel = (uint32_t)(splitmix64(s) >> 32) >> 12;          // 20 bits
        if (sel < 55575u) { for (int k = 0; k < 8; k++) v.v[k] = 0; }
        else if (sel < 55575u + 26214u) { for (int k = 0; k < 8; k++) v.v[k] = 0; v.v[0] = 1; }
    }
    out[i] = Fr::to_mont(v);
}
int gen_scalars(fk_ctx *ctx, Fr *d_out, size_t n, uint64_t seed, int kind) {
    if (!n) return FK_OK;
    hipLaunchKernelGGL(gen_scalars_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n, seed, kind);
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}

// ------------------------------------------------------------------------------------------ density compaction
static constexpr uint32_t CP_BLOCK = 2048;  // elements per block (256 threads x 8)

__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t *dens, size_t n, uint32_t *blk) {
    __shared__ uint32_t sh[256];
    const size_t base = (size_t)blockIdx.x * CP_BLOCK + (size_t)threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) if (base + k < n && dens[base + k]) c++;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 128; off; off >>= 1) { if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) blk[blockIdx.x] = sh[0];
}
// single block exclusive scan of up to 2^16 * 1024 ... entries (each thread strides a contiguous slice)
__global__ __launch_bounds__(1024) void scan_small_kernel(uint32_t *v, uint32_t n, uint32_t *total) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, ipt = (n + 1023) / 1024;
    const uint32_t lo = tid * ipt, hi = lo + ipt < n ? lo + ipt : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi && lo < n; i++) sum += v[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t x = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t i = lo; i < hi && lo < n; i++) { uint32_t x = v[i]; v[i] = run; run += x; }
    if (tid == 1023) *total = part[1023];
}
__global__ __launch_bounds__(256) void compact_scatter_kernel(const Fr *z, const uint8_t *dens, size_t n, const uint32_t *blk, Fr *out) {
    __shared__ uint32_t sh[256];
    const size_t base = (size_t)blockIdx.x * CP_BLOCK + (size_t)threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) if (base + k < n && dens[base + k]) c++;
    sh[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t x = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += x;
        __syncthreads();
    }
    size_t pos = (size_t)blk[blockIdx.x] + sh[threadIdx.x] - c;
    for (int k = 0; k < 8; k++) if (base + k < n && dens[base + k]) out[pos++] = z[base + k];
}

__global__ __launch_bounds__(256) void gather_scalars_kernel(const Fr *z, const uint32_t *idx, size_t n, Fr *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = z[idx[i]];
}

int gather_scalars(fk_ctx *ctx, const Fr *d_z, const uint32_t *d_idx, size_t n, Fr *d_out, hipStream_t st) {
    if (!n) return FK_OK;
    hipLaunchKernelGGL(gather_scalars_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_z, d_idx, n, d_out);
    FK_HIP(ctx, hipGetLastError());
    return FK_OK;
}

int compact_scalars(fk_ctx *ctx, const Fr *d_z, const uint8_t *d_density, size_t n, Fr *d_out, uint64_t *n_out, hipStream_t st) {
    *n_out = 0;
    if (!n) return FK_OK;
    if (!st) st = ctx->stream;
    const uint32_t nb = (uint32_t)((n + CP_BLOCK - 1) / CP_BLOCK);
    FK_HIP(ctx, ctx->scan_tmp.reserve((size_t)nb * 4 + 16));
    uint32_t *blk = ctx->scan_tmp.as<uint32_t>();
    uint32_t *d_total = blk + nb;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(256), 0, st, d_density, n, blk);
    FK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, blk, nb, d_total);
    FK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(compact_scatter_kernel, dim3(nb), dim3(256), 0, st, d_z, d_density, n, blk, d_out);
    FK_HIP(ctx, hipGetLastError());
    uint32_t total = 0;
    FK_HIP(ctx, hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, st));
    FK_HIP(ctx, hipStreamSynchronize(st));
    *n_out = total;
    return FK_OK;
}

}  // namespace fk
