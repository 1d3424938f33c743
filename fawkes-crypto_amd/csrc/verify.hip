// The verifier half: fawkes' `verifier::verify(vk, proof, inputs)`
// (/root/reference/fawkes-crypto/src/backend/bellman_groth16/verifier.rs:75-81 -> bellman's prepare_verifying_key +
// verify_proof) on the host, and for batches on the GPU -- one lane per proof, the lanes of a wave
// running the same Miller loops and the same exponentiation.  Not on the prover's hot path; it exists so that a service
// that proves on the GPU can also check what it ships (and so that the full-size tests do not depend on a Python verifier).
//
// Wire formats are the reference's: the verifying key as fawkes' Borsh `VK` (verifier.rs:46-54: alpha (G1), beta, gamma,
// delta (G2), u32 LE count, ic (G1); every coordinate the canonical little-endian integer, group.rs:16-50), the proof as
// the 256-byte Borsh `Proof` (prover.rs:39-45), the public inputs as `Num<Fr>` (Montgomery limbs), without the leading ONE.
#include "common.hpp"
#include "pairing.hpp"
#include <string.h>

namespace fk {

template <class Fq>
static FK_HD Fq canon_to_mont(const uint8_t *p, bool *ok) {
    Fq v;
    for (int i = 0; i < 8; i++) v.v[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    bool below = false;
    for (int i = 7; i >= 0; i--) {
        if (v.v[i] < FqParams::p(i)) { below = true; break; }
        if (v.v[i] > FqParams::p(i)) break;
    }
    if (!below) *ok = false;                       // Num<Fq>::deserialize: from_uint fails for values >= q
    return Fq::to_mont(v);
}
template <class Fq>
static FK_HD Affine<Fq> g1_from_borsh(const uint8_t *p, bool *ok) { return Affine<Fq>{canon_to_mont<Fq>(p, ok), canon_to_mont<Fq>(p + 32, ok)}; }
template <class Fq>
static FK_HD Affine<Fq2T<Fq>> g2_from_borsh(const uint8_t *p, bool *ok) {
    Affine<Fq2T<Fq>> a;
    a.x.c0 = canon_to_mont<Fq>(p, ok); a.x.c1 = canon_to_mont<Fq>(p + 32, ok);
    a.y.c0 = canon_to_mont<Fq>(p + 64, ok); a.y.c1 = canon_to_mont<Fq>(p + 96, ok);
    return a;
}

// vk: Borsh bytes (alpha 64 | beta 128 | gamma 128 | delta 128 | u32 n_ic | n_ic x 64); inputs: n_ic - 1 Montgomery Fr
// returns 1 accept, 0 reject, -1 malformed encoding
template <class Fq, class FrT>
static FK_HD int verify_one(const uint8_t *vk, uint32_t n_ic, const FrT *inputs, const uint8_t *proof) {
    bool ok = true;
    const Affine<Fq> alpha = g1_from_borsh<Fq>(vk, &ok);
    const Affine<Fq2T<Fq>> beta = g2_from_borsh<Fq>(vk + 64, &ok), gamma = g2_from_borsh<Fq>(vk + 192, &ok), delta = g2_from_borsh<Fq>(vk + 320, &ok);
    const uint8_t *ic = vk + 452;
    Xyzz<Fq> acc = Xyzz<Fq>::from_affine(g1_from_borsh<Fq>(ic, &ok));
    for (uint32_t i = 1; i < n_ic; i++) {
        const FrT k = FrT::from_mont(inputs[i - 1]);
        acc.add(Xyzz<Fq>::mul_scalar(Xyzz<Fq>::from_affine(g1_from_borsh<Fq>(ic + 64 * i, &ok)), k.v));
    }
    const Affine<Fq> A = g1_from_borsh<Fq>(proof, &ok), C = g1_from_borsh<Fq>(proof + 192, &ok);
    const Affine<Fq2T<Fq>> B = g2_from_borsh<Fq>(proof + 64, &ok);
    if (!ok) return -1;
    // The reference decodes the proof unchecked (from_raw_uncompressed_le, group.rs:59-66), so on honest inputs nothing differs;
    // but the Miller loop below assumes B has prime order r (it never meets R = +-Q then) and an off-curve or wrong-subgroup point
    // would make accept / reject unspecified.  Such proofs are REJECTED here: A, C on y^2 = x^3 + 3 (cofactor 1), B on the twist
    // and r * B = identity -- one scalar multiplication per proof, not a hot path.
    {
        const uint32_t bw[8] = FK_G1_B, b0[8] = FK_G2_B0, b1[8] = FK_G2_B1, rw[8] = FK_R_CANON;
        auto cst = [](const uint32_t (&w)[8]) { Fq t; for (int i = 0; i < 8; i++) t.v[i] = w[i]; return t; };
        auto on_g1 = [&](const Affine<Fq> &P) { return P.is_inf() || Fq::sqr(P.y) == Fq::add(Fq::mul(Fq::sqr(P.x), P.x), cst(bw)); };
        using F2 = Fq2T<Fq>;
        if (!on_g1(A) || !on_g1(C)) return 0;
        if (!B.is_inf()) {
            if (F2::sqr(B.y) != F2::add(F2::mul(F2::sqr(B.x), B.x), F2{cst(b0), cst(b1)})) return 0;
            if (!Xyzz<F2>::mul_scalar(Xyzz<F2>::from_affine(B), rw).is_inf()) return 0;
        }
    }
    return groth16_check<Fq>(A, B, C, alpha, beta, gamma, delta, acc.to_affine()) ? 1 : 0;
}

using FrC = Fp<FrParams, false>;
__global__ __launch_bounds__(64) void verify_batch_kernel(const uint8_t *vk, uint32_t n_ic, const FrC *inputs, const uint8_t *proofs, uint32_t count, int8_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = (int8_t)verify_one<FqC, FrC>(vk, n_ic, inputs + (size_t)i * (n_ic - 1), proofs + (size_t)i * FK_PROOF_BYTES);
}

static int vk_check(fk_ctx *ctx, const uint8_t *vk, size_t vk_len, uint32_t n_inputs, uint32_t *n_ic) {
    if (!vk || vk_len < 456) FK_SET_ERR(ctx, FK_ERR_FORMAT, "verify: verifying key truncated");
    uint32_t n; memcpy(&n, vk + 448, 4);
    if (vk_len != 452 + (size_t)n * 64) FK_SET_ERR(ctx, FK_ERR_FORMAT, "verify: verifying key holds %u ic points but is %zu bytes long", n, vk_len);
    // bellman verify_proof: (public_inputs.len() + 1) != pvk.ic.len() -> SynthesisError::MalformedVerifyingKey
    if (n != n_inputs + 1) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "verify: %u public inputs for a key with %u ic points", n_inputs, n);
    *n_ic = n;
    return FK_OK;
}

}  // namespace fk

using namespace fk;

extern "C" {

int fk_verify(fk_ctx *ctx, const uint8_t *vk, size_t vk_len, const uint64_t *inputs, uint32_t n_inputs, const uint8_t proof[FK_PROOF_BYTES], int *accept) { return fk_guard(ctx, [&]() -> int {
    fk_ctx local;                  // host-only routine: usable without a GPU context
    if (!ctx) ctx = &local;
    if (!proof || !accept || (n_inputs && !inputs)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "verify: null argument");
    *accept = 0;
    uint32_t n_ic = 0;
    FK_TRY(vk_check(ctx, vk, vk_len, n_inputs, &n_ic));
    const int r = verify_one<Fq, Fr>(vk, n_ic, (const Fr *)inputs, proof);
    if (r < 0) FK_SET_ERR(ctx, FK_ERR_FORMAT, "verify: a coordinate is not a canonical field element");
    *accept = r;
    return FK_OK;
}); }

int fk_verify_batch_dev(fk_ctx *ctx, const uint8_t *vk, size_t vk_len, const uint64_t *inputs, uint32_t n_inputs, const uint8_t *proofs, uint32_t count,
                        uint8_t *accept) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!proofs || !accept || (n_inputs && !inputs)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "verify: null argument");
    if (!count) return FK_OK;
    uint32_t n_ic = 0;
    FK_TRY(vk_check(ctx, vk, vk_len, n_inputs, &n_ic));
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_b = (size_t)count * n_inputs * 32, pr_b = (size_t)count * FK_PROOF_BYTES;
    const size_t vk_al = (vk_len + 63) & ~(size_t)63, in_al = (in_b + 63) & ~(size_t)63;
    FK_HIP(ctx, ctx->misc.reserve(vk_al + in_al + pr_b + 64));
    FK_HIP(ctx, ctx->stage_d.reserve(count + 64));
    uint8_t *d_vk = ctx->misc.as<uint8_t>(), *d_in = d_vk + vk_al, *d_pr = d_in + in_al;
    int8_t *d_out = ctx->stage_d.as<int8_t>();
    FK_HIP(ctx, hipMemcpyAsync(d_vk, vk, vk_len, hipMemcpyHostToDevice, ctx->stream));
    if (in_b) FK_HIP(ctx, hipMemcpyAsync(d_in, inputs, in_b, hipMemcpyHostToDevice, ctx->stream));
    FK_HIP(ctx, hipMemcpyAsync(d_pr, proofs, pr_b, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(verify_batch_kernel, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, d_vk, n_ic, (const FrC *)d_in, d_pr, count, d_out);
    FK_HIP(ctx, hipGetLastError());
    std::vector<int8_t> res(count);
    FK_HIP(ctx, hipMemcpyAsync(res.data(), d_out, count, hipMemcpyDeviceToHost, ctx->stream));
    FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // A proof that does not decode (a coordinate >= q: upstream fails in `Proof`'s Borsh reader, before verify is reached) is
    // that proof's rejection, not the batch's failure -- one bad submission must not hide the verdicts on the others.
    uint32_t n_bad = 0, first_bad = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (res[i] < 0 && !n_bad++) first_bad = i;
        accept[i] = res[i] > 0 ? 1 : 0;
    }
    if (n_bad) {
        char buf[160];
        snprintf(buf, sizeof buf, "note: %u of %u proofs hold a coordinate that is not a canonical field element (first: proof %u) -- rejected", n_bad, count, first_bad);
        ctx->err = buf;
    }
    return FK_OK;
}); }

}  // extern "C"
