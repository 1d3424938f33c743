/*
 * fawkes_hip.h -- C ABI of libfawkes_hip.so, the MI355X (gfx950) Groth16 proving backend that
 * replaces, for fawkes-crypto, the single call
 *
 *     bellman::groth16::create_random_proof(bcs, &params.0, rng)
 *         /root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80
 *
 * i.e. everything below `prover::prove` (prover.rs:63-90): the Fr NTT/iNTT quotient, the G1/G2
 * Pippenger multi-scalar multiplications and the proof assembly.  The reference has no FFI for
 * this path (it is a generic Rust call into the crate `fawkes-crypto-bellman_ce`); this header is
 * what a Rust `extern "C"` block in a shim crate binds -- see INTEGRATION.md.
 *
 * Data conventions (all little-endian, no padding, no torch types):
 *   Fr / Fq element  32 B = 4 x u64 limbs, MONTGOMERY form, R = 2^256 -- the in-memory image of
 *                    `Num<Fr>` (ff-uint/src/num/mod.rs:21-23; backend/bellman_groth16/mod.rs:105-137).
 *   G1 affine        64 B = x || y, Montgomery LE; the all-zero buffer is the point at infinity --
 *                    `into_raw_uncompressed_le` as used at group.rs:57-66,74-77 (zero: group.rs:55,71).
 *   G2 affine        128 B = x.c0 || x.c1 || y.c0 || y.c1 (group.rs:97-103,114-119).
 *   proof            256 B = fawkes' Borsh `Proof`: a(G1) b(G2) c(G1), every coordinate the CANONICAL
 *                    (non-Montgomery) LE integer (prover.rs:39-45, group.rs:16-21,33-39,
 *                    ff-uint_derive/src/lib.rs:687-693); infinity = zeros.
 *   density maps     one byte per variable, 0/1 -- bellman's DensityTracker bits for a_aux, b_input,
 *                    b_aux (set when the variable is visited in an A- resp. B-side LC).
 *
 * Every function returns FK_OK (0) or an error code and never aborts; fk_last_error() gives text.
 * A context is bound to one GPU and is not thread-safe (one call at a time per context); multi-GPU
 * operation is one process + one context per GPU (see fk_prove_msms / fk_prove_assemble).  Keys (fk_key) and
 * resident constraint systems (fk_r1cs_dev) are read-only device memory once loaded: several contexts on the same
 * GPU, one per host thread, may prove from the same key at the same time -- how a proving service fills the gaps
 * of a single proof (tools/concurrent_probe.py: +5 % proofs/s at 2^25, +21 % at 2^22, +42 % at 2^20 with two).
 */
#ifndef FAWKES_HIP_H
#define FAWKES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fk_ctx fk_ctx;
typedef struct fk_key fk_key;

enum {
    FK_OK = 0,
    FK_ERR_BAD_ARG = 1,
    FK_ERR_DOMAIN_TOO_LARGE = 2,     /* bellman SynthesisError::PolynomialDegreeTooLarge (m > 2^27) */
    FK_ERR_UNEXPECTED_IDENTITY = 3,  /* bellman SynthesisError::UnexpectedIdentity (delta is identity) */
    FK_ERR_HIP = 4,
    FK_ERR_OOM = 5,
    FK_ERR_KEY_MISMATCH = 6,         /* key arrays do not match m / densities (bellman: get_* errors) */
    FK_ERR_FORMAT = 7,               /* malformed file data: std::io::ErrorKind::InvalidData / bellman GroupDecodingError */
    FK_ERR_UNSUPPORTED = 8           /* a run-time dependency is absent (libbrotlidec.so.1 for brotli gate blobs) */
};

#define FK_Z_EQUAL_SPLIT (-1.0)   /* z_frac_lo of every key loader: equal split by shard_index / shard_count */
/* z_frac_lo = FK_Z_WORK_SPLIT: l, a, b_g1, b_g2 laid end to end on a line measured in work (a G2 point counts FK_G2_WORK G1 points) and
 * the line cut into shard_count equal pieces -- a rank holds one or two LARGE pieces instead of 1 / shard_count of each array (b_g1 and
 * b_g2 are then sliced independently: fk_key_shard_info2).  h stays in blocks of the domain.  What fk_multi_* uses from 2 ranks on. */
#define FK_Z_WORK_SPLIT (-2.0)
#define FK_G2_WORK 2.8
/* z_frac_lo = FK_Z_WORK_SPLIT_Q0: the split of the exchange-free schedule ("quotient on rank 0": fk_multi_* on 2, 3, 5, 6, 7 ranks).  Shard 0
 * holds ALL of h -- it evaluates a, b, c, computes the whole quotient and H -- and the work line of l | a | b_g1 | b_g2 is cut so that this fixed
 * work (FK_Q0_HANDICAP G1-point units per domain point) counts towards shard 0's piece; the other shards hold no h at all and never exchange
 * anything but their 384-byte partial sums. */
#define FK_Z_WORK_SPLIT_Q0 (-3.0)
#define FK_Q0_HANDICAP 2.2
#define FK_PROOF_BYTES 256
#define FK_G1_BYTES 64
#define FK_G2_BYTES 128
/* the five MSM results of one prover pass: H, L, A, B1 (G1) and B2 (G2), raw affine LE */
#define FK_MSM_RESULT_BYTES (4 * FK_G1_BYTES + FK_G2_BYTES)

/* ---------------------------------------------------------------- context */
int fk_init(int device_id, fk_ctx **out);
void fk_free(fk_ctx *ctx);
const char *fk_last_error(const fk_ctx *ctx);     /* ctx == NULL: the calling thread's latest context-free call (fk_gates_decode, fk_gates_encode) */
/* 0 = library default.  Pippenger window bits (2..22) used by subsequent MSMs; for tests/tuning. */
/* Releases the scratch the context has grown for the proofs it has run (MSM lanes, transform tables, staging vectors, witness
 * slots); keys and resident constraint systems stay.  Everything is re-allocated on demand.  Not while a proof is submitted. */
int fk_trim(fk_ctx *ctx);
int fk_set_window_bits(fk_ctx *ctx, unsigned c);

/* ---------------------------------------------------------------- device buffers (for resident inputs) */
int fk_dev_alloc(fk_ctx *ctx, size_t bytes, void **dptr);
int fk_dev_free(fk_ctx *ctx, void *dptr);
int fk_upload(fk_ctx *ctx, void *dptr, const void *host, size_t bytes);
int fk_download(fk_ctx *ctx, void *host, const void *dptr, size_t bytes);
/* asynchronous device-to-device copy on the library stream */
int fk_dev_copy(fk_ctx *ctx, void *dst, const void *src, size_t bytes);
int fk_sync(fk_ctx *ctx);
/* the hipStream_t (as void *) of the library's main path -- quotient, SpMV, fk_dq_* are queued there.  A host that runs
 * collectives on streams of its own (RCCL through torch.distributed) orders them against it with stream events
 * (torch.cuda.ExternalStream) instead of fk_sync. */
int fk_stream(fk_ctx *ctx, void **out);
/* pinned host memory (hipHostMalloc): witness buffers handed to fk_prove_r1cs_submit / fk_witness_upload_async */
int fk_host_alloc(fk_ctx *ctx, size_t bytes, void **hptr);
int fk_host_free(fk_ctx *ctx, void *hptr);
/* Two witness slots in device memory, filled on the library's copy stream.  _upload_async returns at once; fk_witness_ptr
 * makes the library's main stream wait for that slot's copy (stream-ordered, the host does not block) and returns the
 * device pointer, to be passed as d_z to the *_dev provers (multi-GPU schedules: every rank uploads the next proof's
 * witness underneath the current proof).  A slot may be refilled once the proof that read it has returned. */
int fk_witness_upload_async(fk_ctx *ctx, int slot, const void *z_host, size_t bytes);
int fk_witness_ptr(fk_ctx *ctx, int slot, void **dptr);
/* The sharded hand-over (N ranks, one witness): a rank uploads only ITS piece over its PCIe link and the ranks exchange the pieces over
 * xGMI with an all-gather, so that the witness crosses PCIe once instead of N times.  fk_multi_prove_r1cs* do this between the ranks of an
 * fk_multi themselves; a rank-per-process host (torch.distributed over RCCL, fawkes-crypto_amd/parallel.py: witness_all_gather) uses:
 *   fk_witness_slot               room for total_bytes in the slot; returns its device pointer (valid until a larger witness is handed
 *                                 over) and the hipStream_t of the copy stream, on which the host issues its collective
 *   fk_witness_upload_part_async  host bytes z_host_part[0, len) -> slot bytes [offset, offset + len), on the copy stream
 *   fk_witness_mark_ready         everything queued on the copy stream so far completes the slot: fk_witness_ptr waits for this point */
int fk_witness_slot(fk_ctx *ctx, int slot, size_t total_bytes, void **dptr, void **copy_stream);
int fk_witness_upload_part_async(fk_ctx *ctx, int slot, const void *z_host_part, size_t offset, size_t len);
int fk_witness_mark_ready(fk_ctx *ctx, int slot);

/* ---------------------------------------------------------------- proving key
 * Replaces the `params.0` argument of prover.rs:80, i.e. bellman's `Parameters` { vk, h, l, a, b_g1,
 * b_g2 } (mod.rs:139).  Uploaded once, stays resident in HBM.  With shard_count > 1 only the
 * [shard_index/shard_count) contiguous slice of each of h, l, a, b_g1, b_g2 is kept (MSM sharding by
 * points across GPUs); the host pointers always describe the FULL arrays. */
typedef struct {
    uint64_t m;              /* evaluation-domain size, power of two, = next_pow2(#gates + num_input) */
    uint32_t num_input;      /* including the constant ONE (cs.rs:111) */
    uint32_t num_aux;
    const uint8_t *alpha_g1, *beta_g1, *delta_g1;   /* 64 B each  */
    const uint8_t *beta_g2, *delta_g2;              /* 128 B each */
    const uint8_t *h;   uint64_t n_h;               /* m - 1 points */
    const uint8_t *l;   uint64_t n_l;               /* num_aux points */
    const uint8_t *a;   uint64_t n_a;               /* num_input + popcount(a_aux) points */
    const uint8_t *b_g1; const uint8_t *b_g2; uint64_t n_b;  /* popcount(b_input)+popcount(b_aux) */
    uint32_t shard_index, shard_count;              /* 0,1 for a single GPU: slice of h */
    /* slice of l, a, b_g1, b_g2: z_frac_lo = FK_Z_EQUAL_SPLIT (any negative value; z_frac_hi ignored) = the same equal
     * split as h.  Otherwise the fractions [z_frac_lo, z_frac_hi) of each array; (0, 0) is the EMPTY slice.  Unequal
     * fractions let the rank that computes the quotient take fewer witness points, possibly none (work-balanced
     * multi-GPU schedule, fawkes-crypto_amd/parallel.py).  A lone shard (shard_count = 1) must hold everything:
     * FK_Z_EQUAL_SPLIT or [0, 1) -- a zero-initialised descriptor is refused with FK_ERR_BAD_ARG. */
    double z_frac_lo, z_frac_hi;
} fk_key_desc;

int fk_key_load(fk_ctx *ctx, const fk_key_desc *desc, fk_key **out);
/* Synthetic key of the same shape (valid curve points, no trapdoor): benchmarking only.  vk points are
 * synthetic too, so proofs made with it do not verify. */
int fk_key_synthetic(fk_ctx *ctx, uint64_t m, uint32_t num_input, uint32_t num_aux, uint64_t n_a,
                     uint64_t n_b, uint64_t seed, uint32_t shard_index, uint32_t shard_count,
                     double z_frac_lo, double z_frac_hi, fk_key **out);
/* out[8] = h_lo, h_hi, l_lo, l_hi, a_lo, a_hi, b_lo, b_hi: the slices this key holds */
int fk_key_shard_info(const fk_key *key, uint64_t out[8]);
/* the same plus b_g2's own slice: out[10] = h, l, a, b_g1, b_g2 ([lo, hi) each) */
int fk_key_shard_info2(const fk_key *key, uint64_t out[10]);
/* Fixed-base precomputation.  The key arrays are fixed bases, and an MI355X has room for more than the key: every
 * loader (fk_key_load, fk_key_load_bellman, fk_setup*, fk_key_synthetic) also derives, HBM permitting, the multiples
 * 2^(offset of window w) * P of each array it holds (W - 1 further copies, W = 11..15), so that the buckets of all
 * Pippenger windows carry the same weights and a multiplication keeps ONE bucket set: one bucket reduction instead of
 * W, and wider windows.  Same group elements, so the proof bytes do not change.  FK_MSM_PRECOMP=0 turns it off; an
 * array whose levels do not fit simply keeps the ordinary path.  out[5] = levels held for h, l, a, b_g1, b_g2 (0 = none). */
int fk_key_precomputed(const fk_key *key, uint32_t out[5]);
/* The loader's choice when the levels do not all fit (2^26, 2^27 on one GPU): the set of arrays with the most accumulation work that fits
 * the HBM left after what a proof will allocate (a knapsack over <= 5 items; b_g1 and b_g2 share a sort, hence both or neither; a G2 point
 * is FK_G2_WORK G1 points of work at twice the bytes).  out[15] = per array h, l, a, b_g1, b_g2: levels held, GiB of levels (negative: the
 * GiB the array WOULD need -- it was left out), estimated ms per proof saved. */
int fk_key_levels_plan(const fk_key *key, double out[15]);
/* (re)derives the fixed-base levels against the HBM that is free now (a key loaded with FK_KEY_NO_LEVELS; or after memory was freed) */
int fk_key_derive_levels(fk_ctx *ctx, fk_key *key);
/* Levels planned EARLY (before a constraint system or anything else large was placed in HBM -- params_io.load_parameters derives them while
 * the gate blob is still being decoded on the host): *bytes = HBM free now - what proofs with this key will still allocate in this context
 * (lane scratch, quotient vectors, witness slots) - 2 GB.  Negative: call fk_key_derive_levels again, it plans against what is free now. */
int fk_key_levels_headroom(fk_ctx *ctx, const fk_key *key, int64_t *bytes);
/* releases the levels (the key itself stays; proofs take the ordinary W-bucket-set path until fk_key_derive_levels is called again) */
int fk_key_drop_levels(fk_ctx *ctx, fk_key *key);
/* what loading the key cost: out[0] = seconds for the arrays themselves (transfer + conversion + the checks of fk_key_load_bellman, or
 * the derivation of fk_setup*), out[1] = seconds for the fixed-base levels */
int fk_key_load_profile(const fk_key *key, double out[2]);
/* Host-only key holding just the vk points fk_prove_assemble needs (no device memory, no GPU).
 * Free with fk_key_free(NULL, key). */
int fk_key_host_vk(const uint8_t *alpha_g1, const uint8_t *beta_g1, const uint8_t *delta_g1,
                   const uint8_t *beta_g2, const uint8_t *delta_g2, fk_key **out);
void fk_key_free(fk_ctx *ctx, fk_key *key);

/* ---------------------------------------------------------------- the prover
 * Inputs are exactly what bellman's ProvingAssignment holds after `synthesize`:
 *   a, b, c    n x 32 B   row evaluations <A_i,z>, <B_i,z>, <C_i,z>, n = #gates + num_input rows
 *   z          (num_input + num_aux) x 32 B   assignment, inputs first (z[0] = ONE)
 *   a_aux_density[num_aux], b_input_density[num_input], b_aux_density[num_aux]
 *   r, s       32 B each  -- the blinding scalars create_random_proof draws from OsRng (prover.rs:78;
 *              sampled limbs are used as the Montgomery representation, App. A.6).  Passing them in
 *              is what makes the proof reproducible: bellman's `create_proof(circuit, params, r, s)`.
 * `*_dev` variants take device pointers for a, b, c, z and the density maps (inputs resident in HBM).
 * a, b, c are consumed as scratch by the _dev variant. */
typedef struct {
    double upload_ms, ntt_ms, msm_h_ms, msm_l_ms, msm_a_ms, msm_b1_ms, msm_b2_ms, assemble_ms, total_ms;
} fk_timings;

int fk_prove(fk_ctx *ctx, const fk_key *key, const uint64_t *a, const uint64_t *b, const uint64_t *c,
             uint64_t n, const uint64_t *z, const uint8_t *a_aux_density, const uint8_t *b_input_density,
             const uint8_t *b_aux_density, const uint64_t r[4], const uint64_t s[4],
             uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);
int fk_prove_dev(fk_ctx *ctx, const fk_key *key, void *d_a, void *d_b, void *d_c, uint64_t n,
                 const void *d_z, const void *d_a_aux_density, const void *d_b_input_density,
                 const void *d_b_aux_density, const uint64_t r[4], const uint64_t s[4],
                 uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);

/* Multi-GPU split of the same computation: each rank runs the quotient and its shard of the five
 * MSMs (fk_prove_msms*), the FK_MSM_RESULT_BYTES partial results are exchanged by the caller
 * (all-gather over RCCL), and any rank folds them into the proof (fk_prove_assemble). */
int fk_prove_msms(fk_ctx *ctx, const fk_key *key, const uint64_t *a, const uint64_t *b, const uint64_t *c,
                  uint64_t n, const uint64_t *z, const uint8_t *a_aux_density, const uint8_t *b_input_density,
                  const uint8_t *b_aux_density, uint8_t out_msms[FK_MSM_RESULT_BYTES], fk_timings *timings);
int fk_prove_msms_dev(fk_ctx *ctx, const fk_key *key, void *d_a, void *d_b, void *d_c, uint64_t n,
                      const void *d_z, const void *d_a_aux_density, const void *d_b_input_density,
                      const void *d_b_aux_density, uint8_t out_msms[FK_MSM_RESULT_BYTES], fk_timings *timings);
/* The two halves of fk_prove_msms_dev, for schedules where one rank computes the quotient and ships h:
 *   fk_prove_msms_z_dev  the witness MSMs L, A, B1, B2 of this key's slices; writes a full
 *                        FK_MSM_RESULT_BYTES record whose H entry is the identity (zeros);
 *   fk_prove_msm_h_dev   H over this key's h slice; d_h_slice points at the (h_hi - h_lo) scalars
 *                        h[h_lo .. h_hi) (Montgomery), result raw affine. */
int fk_prove_msms_z_dev(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux_density,
                        const void *d_b_input_density, const void *d_b_aux_density,
                        uint8_t out_msms[FK_MSM_RESULT_BYTES], fk_timings *timings);
int fk_prove_msm_h_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, uint8_t out[FK_G1_BYTES]);
/* ONE multiplication over one of the key's resident arrays, alone (bench / test micro entry point like fk_msm_g1, SURVEY 8(b);
 * bellman's `multiexp(worker, (bases, 0), FullDensity, scalars)` over that array): `which` = FK_ARRAY_H / _L / _A / _B_G1 / _B_G2,
 * d_scalars holds one Montgomery scalar per point of this key's slice of the array (counts: fk_key_shard_info), and the array's
 * fixed-base levels are used when the key holds them.  Result raw affine: FK_G1_BYTES, FK_G2_BYTES for FK_ARRAY_B_G2. */
#define FK_ARRAY_H 0
#define FK_ARRAY_L 1
#define FK_ARRAY_A 2
#define FK_ARRAY_B_G1 3
#define FK_ARRAY_B_G2 4
int fk_prove_msm_array_dev(fk_ctx *ctx, const fk_key *key, int which, const void *d_scalars, uint8_t *out);
/* Split form for schedules that compute the quotient while the witness multiplications run: _begin queues L, A, B1, B2
 * of this key's slices on the library's MSM streams (nothing is put on the main stream, so fk_quotient_h_dev / fk_dq_*
 * calls issued next overlap with them) and returns at once; _finish adds H over d_h_slice and writes the complete
 * record.  Exactly one _finish per _begin. */
int fk_prove_msms_z_begin_dev(fk_ctx *ctx, const fk_key *key, const void *d_z, const void *d_a_aux,
                              const void *d_b_input_density, const void *d_b_aux_density);
int fk_prove_msms_finish_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, uint8_t out_msms[FK_MSM_RESULT_BYTES]);
/* both at once (the five multiplications are pipelined against each other): the complete FK_MSM_RESULT_BYTES record of
 * this key's slices, given this rank's block of quotient coefficients (distributed quotient, fk_dq_*). */
int fk_prove_msms_hz_dev(fk_ctx *ctx, const fk_key *key, const void *d_h_slice, const void *d_z, const void *d_a_aux,
                         const void *d_b_input_density, const void *d_b_aux_density, uint8_t out_msms[FK_MSM_RESULT_BYTES],
                         fk_timings *timings);
/* ctx may be NULL here (pure host arithmetic). */
int fk_prove_assemble(fk_ctx *ctx, const fk_key *key, const uint8_t *msm_parts, uint32_t n_parts,
                      const uint64_t r[4], const uint64_t s[4], uint8_t out_proof[FK_PROOF_BYTES]);
/* the [lo, hi) slice of an n-element key array (l, a, b_g1, b_g2) that shard `index` of `count` holds */
void fk_shard_range(uint64_t n, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi);
/* the same for the h array (n_h = m - 1 bases): blocks of the evaluation domain, [index*m/count, (index+1)*m/count)
 * clipped to n_h -- the block of quotient coefficients the distributed quotient below leaves on that rank. */
void fk_h_shard_range(uint64_t n_h, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi);
/* the slices of l, a, b_g1, b_g2 ([lo, hi) each: out[8]) that shard `index` of `count` holds under FK_Z_WORK_SPLIT (pure arithmetic:
 * what every key loader applies; a host that shards keys itself calls this) */
void fk_work_shard_ranges(uint64_t n_l, uint64_t n_a, uint64_t n_b, uint32_t index, uint32_t count, uint64_t out[8]);
/* ... under FK_Z_WORK_SPLIT_Q0 (m = the key's domain size: shard 0 carries FK_Q0_HANDICAP * m units of fixed work) */
void fk_work_shard_ranges_q0(uint64_t n_l, uint64_t n_a, uint64_t n_b, uint64_t m, uint32_t index, uint32_t count, uint64_t out[8]);

/* Distributed quotient: bellman's EvaluationDomain pipeline (domain.rs ifft / coset_fft / mul_assign / sub_assign /
 * divide_by_z_on_coset / icoset_fft) over W = 2^log_w GPUs, one process each ("NTT butterfly stages
 * shard across the GPUs").  The m-point transform is cut once between ranks; the library provides the rank-local
 * pieces, the caller moves the data with ONE all-to-all per transform (torch.distributed all_to_all_single = RCCL
 * over xGMI) -- see fawkes-crypto_amd/parallel.py:quotient_distributed for the order.  All buffers are device
 * memory of L = m / W elements (32 B each); chunk p of a buffer is elements [p*L/W, (p+1)*L/W).
 *   fk_dq_gather_dev  local[j] = full[rank + W*j] (zero beyond n): the cyclic slice of a row-evaluation vector
 *   fk_dq_local_dev   stage 0: inverse L-point transform, then * omega_m^(-rank*k)         (ifft, first half)
 *                     stage 1: forward L-point transform                                    (coset_fft, second half)
 *                     stage 2: x := x*xb - xc, then as stage 0                              (icoset_fft, first half)
 *   fk_dq_cross_dev   after an all-to-all (chunk j received from rank j): W-point transforms across the chunks;
 *                     mode 0: ifft second half, * g^i / m, coset_fft first half (result goes into the next all-to-all)
 *                     mode 1: icoset_fft second half, * g^-i / (m Z(g)): quotient coefficients, block-cyclic;
 *                             one more all-to-all leaves rank q with the block h[q*L, (q+1)*L).
 * log_w <= 3, log_m >= 2*log_w.  With log_w = 0 the same calls compute the single-GPU quotient. */
int fk_dq_gather_dev(fk_ctx *ctx, const void *d_full, uint64_t n, uint32_t log_m, uint32_t rank, uint32_t log_w, void *d_local);
int fk_dq_local_dev(fk_ctx *ctx, void *d_x, const void *d_xb, const void *d_xc, uint32_t log_m, uint32_t rank, uint32_t log_w, int stage);
int fk_dq_cross_dev(fk_ctx *ctx, void *d_buf, uint32_t log_m, uint32_t rank, uint32_t log_w, int mode);
/* The six-transform form (what the single-GPU quotient does as well): icoset_fft is linear and undoes coset_fft exactly, so
 * c never has to visit the coset -- h_i = [g^-i / (m Z(g))] ifft(A_c o B_c)_i - [1 / Z(g)] ifft(c)_i, the same field elements.
 *   fk_dq_cross_dev mode 2   ifft second half of c only, * 1 / (m Z(g)): c's scaled coefficients, block-cyclic, stay put
 *   fk_dq_local_dev stage 3  x := x * xb, then as stage 0
 *   fk_dq_cross_sub_dev      mode 1, then - d_sub (same block-cyclic layout)
 * One all-to-all and one rank-local transform fewer per proof: 7 exchanges instead of 8 (parallel.quotient_distributed). */
int fk_dq_cross_sub_dev(fk_ctx *ctx, void *d_buf, const void *d_sub, uint32_t log_m, uint32_t rank, uint32_t log_w);

/* ---------------------------------------------------------------- building blocks (tests / benches)
 * bellman_ce::domain::EvaluationDomain pieces. */
int fk_fr_mul_batch(fk_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
/* in-place natural-order NTT of 2^log_n elements; inverse: omega^-1 and 1/n; coset: bellman's
 * coset_fft / icoset_fft (multiplicative generator 7). */
int fk_ntt(fk_ctx *ctx, uint64_t *data, uint32_t log_n, int inverse, int coset);
int fk_ntt_dev(fk_ctx *ctx, void *d_data, uint32_t log_n, int inverse, int coset);
/* h = (A*B - C)/Z coefficients: out has m-1 elements, m = next_pow2(n). */
int fk_quotient_h(fk_ctx *ctx, const uint64_t *a, const uint64_t *b, const uint64_t *c, uint64_t n,
                  uint64_t *h_out);
int fk_quotient_h_dev(fk_ctx *ctx, void *d_a, void *d_b, void *d_c, uint64_t n, void *d_h_out /* m x 32 B */);
/* bellman_ce::multiexp (App. A.3): sum_i scalars[i] * bases[i]; scalars Montgomery Fr; out raw affine. */
int fk_msm_g1(fk_ctx *ctx, const uint8_t *bases, const uint64_t *scalars, size_t n, uint8_t out[FK_G1_BYTES]);
int fk_msm_g2(fk_ctx *ctx, const uint8_t *bases, const uint64_t *scalars, size_t n, uint8_t out[FK_G2_BYTES]);
int fk_msm_g1_dev(fk_ctx *ctx, const void *d_bases, const void *d_scalars, size_t n, uint8_t out[FK_G1_BYTES]);
int fk_msm_g2_dev(fk_ctx *ctx, const void *d_bases, const void *d_scalars, size_t n, uint8_t out[FK_G2_BYTES]);
/* n valid pseudo-random curve points written to device memory (bench/test input generator) */
int fk_gen_points_g1_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed);
int fk_gen_points_g2_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed);
/* n pseudo-random Montgomery Fr elements; kind 0 = uniform, 1 = witness-like (half in {0,1}), 2 = the benchmark witness's mix (5.3 % zeros,
 * 2.5 % ones, the rest dense and pairwise distinct) */
int fk_gen_scalars_dev(fk_ctx *ctx, void *d_out, size_t n, uint64_t seed, int kind);

/* ---------------------------------------------------------------- synthesis (host side of the boundary)
 * ProvingAssignment::enforce/eval restated (App. A.1): evaluates a CSR R1CS on the assignment and
 * produces a, b, c and the density maps, i.e. what backend/bellman_groth16/mod.rs:92-99 feeds bellman.
 * Variables: Input(i) -> i, Aux(j) -> num_input + j (circuit/r1cs/cs.rs:255-268).  ctx may be NULL. */
typedef struct {
    uint32_t num_input, num_aux;
    uint64_t num_gates;
    const uint64_t *a_ptr; const uint32_t *a_col; const uint64_t *a_val;   /* ptr[num_gates+1], col/val[nnz]; val == NULL: all ONE */
    const uint64_t *b_ptr; const uint32_t *b_col; const uint64_t *b_val;
    const uint64_t *c_ptr; const uint32_t *c_col; const uint64_t *c_val;
} fk_r1cs;

int fk_synthesize(fk_ctx *ctx, const fk_r1cs *cs, const uint64_t *z, uint64_t *a, uint64_t *b, uint64_t *c,
                  uint8_t *a_aux_density, uint8_t *b_input_density, uint8_t *b_aux_density);

/* ---------------------------------------------------------------- device-resident constraint system
 * (SURVEY section 8f row 1).  fk_r1cs_load uploads the CSR matrices once (coefficients dictionary-encoded)
 * and derives the structural density maps; afterwards only the witness vector crosses the boundary:
 * fk_prove_r1cs = SpMV (a = Az, b = Bz, c = Cz + the per-input rows) -> quotient -> MSMs -> assembly,
 * i.e. the whole of `create_proof` behind prover.rs:80 including bellman's `synthesize` evaluation
 * (mod.rs:92-99).  z: num_input + num_aux Montgomery elements, inputs first, z[0] = ONE. */
typedef struct fk_r1cs_dev fk_r1cs_dev;
int fk_r1cs_load(fk_ctx *ctx, const fk_r1cs *cs, fk_r1cs_dev **out);
/* Batch circuits (BASELINE configs[2]: one R1CS holding 4096 eddsa verifiers): `instance` is ONE copy of the gadget and
 * the resident system stands for `copies` of it -- the gates of copy j are rows [j*G, (j+1)*G), the constant ONE is shared,
 * the variables are ordered ONE, copy 0's inputs, copy 1's inputs, ..., copy 0's aux, copy 1's aux, ... -- without
 * materialising copies x nnz terms (2.2e9 for that batch): the kernel maps a copy's variables by arithmetic.  The
 * result is indistinguishable from fk_r1cs_load of the explicitly replicated system (tests/test_gpu_tiled.py). */
int fk_r1cs_load_tiled(fk_ctx *ctx, const fk_r1cs *instance, uint32_t copies, fk_r1cs_dev **out);
/* the same with the coefficients already dictionary-coded by the caller (cs->*_val ignored): x_cidx[i] indexes `table`
 * (n_table Montgomery values, table[0] must be ONE) -- 8 bytes per term on the host side too, for systems of 10^9 terms */
int fk_r1cs_load_coded(fk_ctx *ctx, const fk_r1cs *cs, const uint32_t *a_cidx, const uint32_t *b_cidx, const uint32_t *c_cidx,
                       const uint64_t *table, uint64_t n_table, fk_r1cs_dev **out);
void fk_r1cs_free(fk_ctx *ctx, fk_r1cs_dev *r1cs);
/* out[8] = rows, nnz(A), nnz(B), nnz(C), distinct coefficients, points the a query needs, points the b query needs,
 * variables (num_input + num_aux: the length of the witness vector the prove calls read) */
int fk_r1cs_info(const fk_r1cs_dev *r1cs, uint64_t out[8]);
/* The row windows of the chunked witness hand-over (fk_prove_r1cs, round 5).  An explicit system of >= 65536 gates whose three matrices
 * all have the length-class lists is cut at load into *n_windows (8; FK_SPMV_WINDOWS) windows of consecutive gates: rows[j] .. rows[j + 1]
 * are the gates of window j and need[j] the number of leading witness elements the windows 0 .. j read (need[last] = all).  fk_prove_r1cs
 * then uploads the caller's witness in those pieces and evaluates window j as soon as piece j is on the device, so the 19 ms upload of the
 * benchmark's witness hides the evaluation instead of preceding it.  *n_windows = 0: no windows (small, tiled or unbinned system) -- the
 * witness is uploaded whole, as before.  FK_PROVE_CHUNKED_UPLOAD=0 disables the chunked path.  Either way the proof bytes are the same. */
int fk_r1cs_windows(const fk_r1cs_dev *r1cs, uint32_t *n_windows, uint64_t rows[17], uint64_t need[16]);
/* device pointers of the structural density maps: a_aux[num_aux], b_input[num_input], b_aux[num_aux] */
int fk_r1cs_density_ptrs(const fk_r1cs_dev *r1cs, const void *out[3]);
/* a, b, c <- A z, B z, C z on the device (arrays sized for next_pow2(rows) elements, `rows` written) */
int fk_r1cs_eval_dev(fk_ctx *ctx, const fk_r1cs_dev *r1cs, const void *d_z, void *d_a, void *d_b, void *d_c);
/* the cyclic row slice one of 2^log_w ranks needs: local[j] = row (rank + j * 2^log_w) of A z, B z, C z, zero behind the last
 * row -- what fk_dq_gather_dev cuts out of the full vectors, at 1 / 2^log_w of the work and without the three m-element
 * vectors.  d_a, d_b, d_c: 2^(log_m - log_w) elements each. */
int fk_r1cs_eval_slice_dev(fk_ctx *ctx, const fk_r1cs_dev *r1cs, const void *d_z, uint32_t log_m, uint32_t rank, uint32_t log_w,
                           void *d_a, void *d_b, void *d_c);
int fk_prove_r1cs(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r1cs, const uint64_t *z,
                  const uint64_t r[4], const uint64_t s[4], uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);
/* multi-GPU counterpart of fk_prove_msms_hz_dev for a resident constraint system (its A / B query index lists replace
 * the per-proof density compaction) */
int fk_prove_msms_hz_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r1cs, const void *d_h_slice, const void *d_z,
                              uint8_t out_msms[FK_MSM_RESULT_BYTES]);
/* ... and of fk_prove_msms_z_begin_dev (finish with fk_prove_msms_finish_dev) */
int fk_prove_msms_z_begin_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r1cs, const void *d_z);
int fk_prove_r1cs_dev(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r1cs, const void *d_z,
                      const uint64_t r[4], const uint64_t s[4], uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);
/* The same proof, pipelined over the host boundary.  The reference produces the witness on the host for every proof
 * (prover.rs:69-76: the circuit closure fills WitnessCS, then prover.rs:80 consumes it), so a proving loop hands over
 * (num_input + num_aux) * 32 bytes per proof -- 1 GiB at 2^25 variables.  _submit starts the host-to-device copy into one
 * of two witness slots on a copy stream and returns at once; _wait computes the proof of that ticket.  With
 * submit(k+1) issued before wait(k) the upload of the next witness runs underneath the current proof (with a ticket already
 * outstanding the copy is queued by THAT proof's run, behind its memory-bound front, so that it overlaps with transforms and
 * accumulations rather than with sorts).  At most two
 * tickets are outstanding; z must stay valid until the matching _wait returns and should be pinned memory
 * (fk_host_alloc) -- pageable memory works but is staged by the runtime and does not overlap.
 * With a second ticket of the same key and system outstanding, _wait(k) also queues the FRONT of proof k+1 -- the evaluation of
 * a, b, c and the witness multiplications' sorts -- behind proof k's last accumulation, where proof k only has latency-bound
 * tails left (sizes that run the sorts-first schedule, 2^25 and up: "early front", csrc/spmv.hip); _wait(k+1) picks it up.
 * Between a _submit and its _wait only _submit / _wait may be called on the context. */
int fk_prove_r1cs_submit(fk_ctx *ctx, const fk_key *key, const fk_r1cs_dev *r1cs, const uint64_t *z,
                         const uint64_t r[4], const uint64_t s[4], int *ticket);
int fk_prove_r1cs_wait(fk_ctx *ctx, int ticket, uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);

/* ---------------------------------------------------------------- the circuit half of a `Parameters` file
 * fawkes stores the constraint system as `Parameters.2`: brotli(concatenated Borsh gates) (setup.rs:25-32) and replays it
 * for every proof through WitnessCS::get_gate_iterator / GateStreamedIterator (cs.rs:184-223, 243-245).  Here it is decoded
 * ONCE, streamed and native: blob -> CSR with dictionary-coded coefficients (host) -> resident constraint system.
 * FK_GATES_BROTLI binds the system's libbrotlidec.so.1 at run time (FK_ERR_UNSUPPORTED if absent); FK_GATES_RAW is the bare
 * gate stream.  Malformed input (bad tag, index out of range, coefficient >= r, truncation, trailing bytes) is FK_ERR_FORMAT.
 * num_gates is `Parameters.1`; num_input / num_aux come from the key (ic and l lengths).  ctx may be NULL for fk_gates_decode. */
#define FK_GATES_RAW 0
#define FK_GATES_BROTLI 1
typedef struct fk_gates fk_gates;
int fk_gates_decode(fk_ctx *ctx, const uint8_t *blob, size_t len, int format, uint32_t num_gates, uint32_t num_input,
                    uint32_t num_aux, fk_gates **out);
void fk_gates_free(fk_gates *gates);
/* out[8] = num_gates, nnz(A), nnz(B), nnz(C), distinct coefficients, decoded bytes, num_input, num_aux */
int fk_gates_info(const fk_gates *gates, uint64_t out[8]);
/* matrix mtx (0 = A, 1 = B, 2 = C) as the arrays of an fk_r1cs: ptr[num_gates + 1], col[nnz] and (if non-NULL) val[nnz x 4] */
int fk_gates_export(const fk_gates *gates, int mtx, uint64_t *ptr, uint32_t *col, uint64_t *val);
/* the resident constraint system of the decoded stream (as fk_r1cs_load; the dictionary and the structural density flags the
 * decoder derived while parsing are taken over as they are) */
int fk_r1cs_load_gates(fk_ctx *ctx, const fk_gates *gates, fk_r1cs_dev **out);
/* how the decoding went: out[8] = wall seconds, seconds inside the decompressor, seconds the decoding thread waited for a free
 * parsing thread, parsing seconds summed over the threads, seconds renumbering the dictionary, parsing threads, blocks, blob bytes.
 * (The decompressor is one serial bit stream -- the floor; the parsing runs beside it on FK_HOST_THREADS threads, default: the
 * cores this process may use.) */
int fk_gates_profile(const fk_gates *gates, double out[8]);
/* The writer's side (setup.rs:25-32: `Parameters.2` = brotli(quality 9, lgwin 22) over Gate::serialize of every gate, cs.rs:184-191):
 * the gate blob of `copies` copies of `cs` (fk_r1cs_load_tiled's variable order; 1 = the system itself), FK_GATES_BROTLI through the
 * system's libbrotlienc.so.1 (FK_ERR_UNSUPPORTED if absent; quality 0 .. 11, lgwin 10 .. 24 -- any setting decodes to the same stream; 2 .. 9 at the same speed, 0 and 1 slower) or
 * FK_GATES_RAW (the bare stream; quality / lgwin ignored).  The stream is formatted by FK_HOST_THREADS threads and never exists as
 * a whole: the benchmark's 61 GB stream becomes a 2.8 GB blob at quality 1, 1.35 GB at quality 2 (which decodes 1.5 x faster, like a quality-9 blob).  ctx may be NULL. */
typedef struct fk_blob fk_blob;
int fk_gates_encode(fk_ctx *ctx, const fk_r1cs *cs, uint32_t copies, int format, int quality, int lgwin, fk_blob **out);
int fk_blob_data(const fk_blob *blob, const uint8_t **data, size_t *len);
/* out[4] = wall seconds, seconds inside the compressor, stream bytes, blob bytes */
int fk_blob_profile(const fk_blob *blob, double out[4]);
void fk_blob_free(fk_blob *blob);

/* ---------------------------------------------------------------- key generation on the GPU
 * (SURVEY section 8f row 4): bellman's generate_parameters (reached from setup.rs:20) for EXPLICIT toxic
 * waste tau, alpha, beta, gamma, delta (Montgomery Fr) and the standard BN254 generators.  Produces a
 * resident proving key; with shard_count > 1 only the requested shard of every array is derived (W ranks do 1/W of the fixed-base work
 * each); shard 0 of 1 with z_frac_lo = FK_Z_EQUAL_SPLIT = whole key.  vk_out: six 128-byte slots alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1,
 * delta_g2 (G1 points use the first 64 bytes); ic_out: num_input x 64 bytes.  For tests and benchmarks:
 * a real deployment runs an MPC ceremony, never a setup with known toxic waste. */
int fk_setup(fk_ctx *ctx, const fk_r1cs *cs, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4],
             const uint64_t gamma[4], const uint64_t delta[4], uint32_t shard_index, uint32_t shard_count,
             double z_frac_lo, double z_frac_hi, fk_key **out_key, uint8_t vk_out[6 * 128], uint8_t *ic_out);
/* the key of `copies` instances as one system (layout of fk_r1cs_load_tiled); ic_out: (1 + copies*(num_input-1)) x 64 bytes */
int fk_setup_tiled(fk_ctx *ctx, const fk_r1cs *instance, uint32_t copies, const uint64_t tau[4], const uint64_t alpha[4],
                   const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4], uint32_t shard_index,
                   uint32_t shard_count, double z_frac_lo, double z_frac_hi, fk_key **out_key, uint8_t vk_out[6 * 128],
                   uint8_t *ic_out);
/* copies this key's slice of one array to the host; which: 0 = h, 1 = l, 2 = a, 3 = b_g1, 4 = b_g2 */
int fk_key_download(fk_ctx *ctx, const fk_key *key, int which, void *host, size_t host_bytes);

/* ---------------------------------------------------------------- bellman key files
 * Loads the bellman part of a fawkes `Parameters` file (what `self.0.write(writer)` emits at mod.rs:156, i.e.
 * bellman_ce's Parameters::write: big-endian uncompressed points, SURVEY Appendix B.2 -- layout restated from
 * the upstream crate, NOT verifiable against the reference here) into the resident device layout; the
 * big-endian -> Montgomery conversion runs on the GPU.  gamma_g2_out (128 B, may be NULL) and ic_out (ic_cap x 64 B,
 * may be NULL) receive the verifier-only parts as raw Montgomery LE.  The fawkes header in front of it
 * (num_gates, gate blob, const_tracker) is parsed by the host layer (fawkes-crypto_amd/params_io.py), the gate blob by
 * fk_gates_decode.  `flags` carries the two arguments of `Parameters::read(reader, disallow_points_at_infinity, checked)`
 * (mod.rs:159): FK_KEY_NO_INFINITY | FK_KEY_CHECKED.  Whatever the flags, coordinates must be canonical (< q), the
 * compression bit clear and an infinity encoding clean -- what the unchecked decoding of pairing_ce enforces too.  Any
 * violation is FK_ERR_FORMAT (bellman: io::Error from GroupDecodingError) with the counts in fk_last_error. */
#define FK_KEY_CHECKED 1u       /* `checked`: every point on its curve, G2 points in the order-r subgroup (checked on the GPU) */
#define FK_KEY_NO_INFINITY 2u   /* `disallow_points_at_infinity`: no identity point in h, l, a, b_g1, b_g2 */
#define FK_KEY_NO_LEVELS 4u     /* do not derive the fixed-base levels now: the caller calls fk_key_derive_levels once everything else it
                                 * wants in HBM (a resident constraint system) has been placed */
int fk_key_load_bellman(fk_ctx *ctx, const uint8_t *buf, size_t len, uint32_t flags, uint32_t shard_index, uint32_t shard_count,
                        double z_frac_lo, double z_frac_hi, fk_key **out, uint8_t *gamma_g2_out, uint8_t *ic_out,
                        uint32_t ic_cap, uint32_t *n_ic);
/* bellman `Parameters::write` (mod.rs:156) of a WHOLE key resident in HBM -- the counterpart of fk_key_load_bellman: vk, then h, l, a, b_g1
 * (G1) and b_g2 (G2), each a u32 BE count + uncompressed big-endian points, converted on the GPU.  gamma_g2 (128 B raw) and ic (n_ic x 64 B
 * raw, n_ic = num_input) are the caller's (a proving key does not hold them: fk_setup* / fk_key_load_bellman return them).  out == NULL:
 * only *needed is set.  fawkes' own header (gate count, gate blob, const-tracker bits, mod.rs:150-155) is written by the host. */
int fk_key_write_bellman(fk_ctx *ctx, const fk_key *key, const uint8_t *gamma_g2, const uint8_t *ic, uint32_t n_ic, uint8_t *out, size_t cap,
                         size_t *needed);
/* alpha_g1, beta_g1, delta_g1 (64 B each) then beta_g2, delta_g2 (128 B each), raw Montgomery LE */
int fk_key_vk(const fk_key *key, uint8_t out[3 * 64 + 2 * 128]);
/* out[8] = m, num_input, num_aux, n_h, n_l, n_a, n_b, shard_count */
int fk_key_counts(const fk_key *key, uint64_t out[8]);

/* ---------------------------------------------------------------- multi-GPU prover: one call, N GPUs of one node
 * `fk_init(int n_devices, const int *device_ids)`.  The reference's entry is ONE call --
 * `prove(params, pub, sec, circuit)` (prover.rs:63-90) -- and so is the sharded form: one process, one library context and
 * one worker thread per GPU, every exchange inside the library (csrc/multi.hip).  Rank g (= position in device_ids) keeps
 * shard g of every key array, evaluates the rows t = g (mod N) of a, b, c, computes 1/N of the quotient (every transform is
 * cut once between the ranks; its all-to-all is device-to-device DMA over xGMI -- hipMemcpyPeerAsync pulls on an exchange
 * stream per rank, ordered with HIP events against the kernels on both sides, seven per proof) and 1/N of each of the five
 * multi-scalar multiplications; the N 384-byte partial results are folded on the host ("all-reduce of the partial sums":
 * no collective library has an elliptic-curve reduction operator).  The WITNESS crosses PCIe once: rank g uploads the piece
 * z[g * C, (g + 1) * C), C = ceil(variables / N), over its own link and every rank collects the other pieces from its peers over
 * xGMI (an all-gather: peer DMA on the ranks' copy streams, or one grouped ncclAllGather with FK_MULTI_TRANSPORT=rccl) underneath the
 * proof before; FK_MULTI_WITNESS=whole restores rounds 3-4 (every rank uploads all of z).  N = 1, 2, 4, 8 use this schedule; any other N up to 64
 * shards the multiplications only (every rank computes the whole quotient).  Device ids may repeat: several ranks then share
 * a GPU (how a one-GPU box tests the path).  The proof bytes do not depend on N.
 * An fk_multi is not thread-safe (one call at a time); keys and constraint systems loaded through it belong to it.
 * FK_OVERLAP_WITNESS=1 / 0: begin the witness multiplications before the quotient (default from 4 ranks on);
 * FK_MULTI_HOST_EVENTS=1: wait for a peer GPU's event on the host instead of in the stream (diagnosis);
 * FK_MULTI_TRANSPORT=rccl: the seven exchanges as grouped ncclSend / ncclRecv over one RCCL communicator per rank (librccl.so.1
 * bound with dlopen; distinct devices only; falls back to the peer copies with a note in fk_multi_last_error);
 * FK_MULTI_FORCE_EXCHANGE=1: run the distributed schedule even with one rank (test aid). */
typedef struct fk_multi fk_multi;
typedef struct fk_multi_key fk_multi_key;
typedef struct fk_multi_r1cs fk_multi_r1cs;
int fk_init_devices(int n_devices, const int *device_ids, fk_multi **out);
void fk_multi_free(fk_multi *multi);
const char *fk_multi_last_error(const fk_multi *multi);
int fk_multi_size(const fk_multi *multi);
/* "peer-dma" (hipMemcpyPeerAsync pulls, the default) or "rccl" */
const char *fk_multi_transport(const fk_multi *multi);
/* What fk_init_devices found when it asked for direct access between the ranks' devices (hipDeviceCanAccessPeer /
 * hipDeviceEnablePeerAccess), per ORDERED pair: out[i * N + j] describes copies INTO rank i's device FROM rank j's.  A pair that is not
 * FK_PEER_DIRECT still works -- the runtime stages its copies -- and fk_multi_last_error carries a note counting such pairs. */
#define FK_PEER_SELF 0     /* both ranks name the same device */
#define FK_PEER_DIRECT 1   /* peer access granted (or already on): hipMemcpyPeerAsync is device-to-device DMA over xGMI */
#define FK_PEER_STAGED 2   /* hipDeviceCanAccessPeer says no: copies are staged by the runtime */
#define FK_PEER_REFUSED 3  /* hipDeviceEnablePeerAccess refused: copies are staged by the runtime */
int fk_multi_topology(const fk_multi *multi, int32_t *out);
/* First contact with a node, before any key is built: for every ordered pair of ranks (i != j) a `bytes`-byte pull into rank i's device
 * from rank j's on rank i's exchange stream, ordered behind an event recorded on rank j's main stream (the cross-device wait every
 * exchange of a proof uses), timed with HIP events on that stream and compared byte for byte on the host; one pair at a time.
 * gbps[i * N + j]: GB/s of the pull; status[i * N + j]: FK_OK or the failing code (the message of the last failure: fk_multi_last_error).
 * If the in-stream wait on another device's event fails and the host-side wait works, the context switches ITSELF to host-side waits
 * (what FK_MULTI_HOST_EVENTS=1 selects), sets *host_events_out = 1 (may be NULL) and leaves a note.  `bytes`: a multiple of 8, 64 .. 2^32.
 * Returns FK_OK when every pair passed. */
int fk_multi_preflight(fk_multi *multi, size_t bytes, double *gbps, int32_t *status, int *host_events_out);
/* rank `rank`'s single-GPU context (statistics, calibration, building blocks); owned by the fk_multi */
fk_ctx *fk_multi_ctx(fk_multi *multi, int rank);
int fk_multi_sync(fk_multi *multi);
/* bytes the latest witness hand-over moved, summed over the ranks: out[0] host -> device (PCIe), out[1] device -> device (all-gather) */
int fk_multi_witness_traffic(const fk_multi *multi, uint64_t out[2]);
/* the key loaders of the single-GPU interface, shard g of N on rank g (desc->shard_index / shard_count / z_frac_* are ignored) */
int fk_multi_key_load(fk_multi *multi, const fk_key_desc *desc, fk_multi_key **out);
int fk_multi_key_load_bellman(fk_multi *multi, const uint8_t *buf, size_t len, uint32_t flags, fk_multi_key **out, uint8_t *gamma_g2_out,
                              uint8_t *ic_out, uint32_t ic_cap, uint32_t *n_ic);
int fk_multi_setup(fk_multi *multi, const fk_r1cs *cs, const uint64_t tau[4], const uint64_t alpha[4], const uint64_t beta[4],
                   const uint64_t gamma[4], const uint64_t delta[4], fk_multi_key **out, uint8_t vk_out[6 * 128], uint8_t *ic_out);
int fk_multi_setup_tiled(fk_multi *multi, const fk_r1cs *instance, uint32_t copies, const uint64_t tau[4], const uint64_t alpha[4],
                         const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta[4], fk_multi_key **out,
                         uint8_t vk_out[6 * 128], uint8_t *ic_out);
void fk_multi_key_free(fk_multi *multi, fk_multi_key *key);
const fk_key *fk_multi_key_shard(const fk_multi_key *key, int rank);
/* the resident constraint system, one replica per GPU */
int fk_multi_r1cs_load(fk_multi *multi, const fk_r1cs *cs, fk_multi_r1cs **out);
int fk_multi_r1cs_load_tiled(fk_multi *multi, const fk_r1cs *instance, uint32_t copies, fk_multi_r1cs **out);
int fk_multi_r1cs_load_gates(fk_multi *multi, const fk_gates *gates, fk_multi_r1cs **out);
void fk_multi_r1cs_free(fk_multi *multi, fk_multi_r1cs *r1cs);
const fk_r1cs_dev *fk_multi_r1cs_replica(const fk_multi_r1cs *r1cs, int rank);
/* witness in (host memory; every GPU uploads it over its own PCIe link) -> proof out: the multi-GPU fk_prove_r1cs, and its
 * two-slot pipelined form (the upload of proof k+1 runs underneath proof k; z must stay valid until the matching _wait
 * returns and should be pinned -- fk_host_alloc on fk_multi_ctx(multi, 0)) */
int fk_multi_prove_r1cs(fk_multi *multi, const fk_multi_key *key, const fk_multi_r1cs *r1cs, const uint64_t *z, const uint64_t r[4],
                        const uint64_t s[4], uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);
int fk_multi_prove_r1cs_submit(fk_multi *multi, const fk_multi_key *key, const fk_multi_r1cs *r1cs, const uint64_t *z,
                               const uint64_t r[4], const uint64_t s[4], int *ticket);
int fk_multi_prove_r1cs_wait(fk_multi *multi, int ticket, uint8_t out_proof[FK_PROOF_BYTES], fk_timings *timings);

/* ---------------------------------------------------------------- verifier
 * `verifier::verify(vk, proof, inputs)` (verifier.rs:75-81 -> bellman's verify_proof).  vk: fawkes' Borsh `VK`
 * (verifier.rs:46-54: alpha G1 | beta, gamma, delta G2 | u32 LE count | ic G1 points; canonical little-endian coordinates);
 * proof: the 256-byte Borsh `Proof`; inputs: n_inputs Montgomery Fr, the public inputs without the leading ONE.
 * *accept = 1 / 0.  n_inputs + 1 != #ic is FK_ERR_KEY_MISMATCH (bellman: MalformedVerifyingKey), a non-canonical
 * coordinate FK_ERR_FORMAT.  fk_verify runs on the host (ctx may be NULL, ~25 ms); fk_verify_batch_dev checks `count`
 * proofs of the same key on the GPU, one lane per proof (inputs: count x n_inputs, proofs: count x 256 B, host memory);
 * accept[i] = 1 / 0, and a proof with a non-canonical coordinate is REJECTED (0) like any other bad proof -- the call
 * still returns FK_OK with the verdicts on the rest (fk_last_error then holds a note naming the first such proof). */
int fk_verify(fk_ctx *ctx, const uint8_t *vk, size_t vk_len, const uint64_t *inputs, uint32_t n_inputs,
              const uint8_t proof[FK_PROOF_BYTES], int *accept);
int fk_verify_batch_dev(fk_ctx *ctx, const uint8_t *vk, size_t vk_len, const uint64_t *inputs, uint32_t n_inputs,
                        const uint8_t *proofs, uint32_t count, uint8_t *accept);

/* Kernel timing measured with HIP events on the library's stream since the last reset, summed over
 * launches.  which: 0 = msm_accumulate_kernel<Fq> (G1 bucket accumulation; units = points per launch),
 * 1 = msm_accumulate_kernel<Fq2> (G2), 2 = ntt_pass_kernel (units = elements per pass); 3 / 4 = the same kernels as 0 / 1
 * with units = the mixed point additions they performed (the work unit of the VALU roofline: 10 modular products each
 * in G1), counted on the device from the sorted bucket sizes; 5 / 6 = the same kernels as 0 / 1 with ms = the UNION of the
 * launches' intervals (launches of one kernel that run side by side on different lanes each span the whole phase: the union
 * is the time the kernel took, the sum counts it once per launch). */
/* Tracing: with FK_ROCTX=1 in the environment the library brackets its phases with roctx ranges (libroctx64.so.4, bound with
 * dlopen on first use) -- key loads, the gate decoder, set-up, and inside a proof the queueing of the front, of the quotient, of the next proof's
 * early front and the wait for the five multiplications -- so that `rocprofv3 --marker-trace --kernel-trace` shows which call queued which
 * kernels.  Returns 1 when ranges are being emitted, 0 otherwise (unset, or the library is absent: never an error). */
int fk_roctx_active(void);
int fk_stats_reset(fk_ctx *ctx);
/* Live calibration of the VALU ceilings the measurement quotes (a few milliseconds): out[0] = v_mad_u64_u32 lane-operations
 * per second (the 32 x 32 -> 64-bit multiply-accumulate every Montgomery product is made of), out[1] = Montgomery products
 * per second of the library's multiplier running alone in registers. */
int fk_calibrate(fk_ctx *ctx, double out[2]);
int fk_stats_get(fk_ctx *ctx, int which, double *ms, uint64_t *launches, uint64_t *units);

#ifdef __cplusplus
}
#endif
#endif /* FAWKES_HIP_H */
