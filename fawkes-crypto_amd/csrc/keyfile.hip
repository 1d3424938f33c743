// Loading a bellman-serialised proving key straight into the device layout.
//
// fawkes' `Parameters::write` (/root/reference/fawkes-crypto/src/backend/bellman_groth16/mod.rs:150-157) emits
// its own header (num_gates, brotli gate blob, const_tracker bit vector -- parsed on the host side, see
// fawkes-crypto_amd/params_io.py) followed by bellman's `Parameters::write`.  That second part lives in the
// un-vendored crate fawkes-crypto-bellman_ce 0.3.5 / pairing_ce 0.18.1, so its layout is restated from the
// upstream sources and could NOT be checked against a file written by the reference:
//   vk:  alpha_g1, beta_g1 (G1), beta_g2, gamma_g2 (G2), delta_g1 (G1), delta_g2 (G2), u32 BE count, ic[] (G1)
//   then h, l, a, b_g1 (G1) and b_g2 (G2), each: u32 BE count + points
//   G1 uncompressed = x || y, 32-byte BIG-endian canonical integers; G2 uncompressed = x.c1 || x.c0 || y.c1 || y.c0;
//   point at infinity = first byte 0x40, rest zero.
// The device kernel turns every 32-byte big-endian coordinate into the Montgomery little-endian limbs the
// kernels use (byte reversal + one Montgomery multiplication by R^2) and reorders G2 to c0 || c1.
#include "common.hpp"
#include <chrono>
#include <string.h>

namespace fk {

// error counters of one load (device): what pairing_ce's GroupDecodingError distinguishes
struct KeyErr { uint32_t flag, range, inf_rest, curve, subgroup, infinity; };

// 32 big-endian bytes -> canonical limbs; false when the integer is not below q (CoordinateDecodingError: "not in field")
static __device__ __forceinline__ bool be32_to_mont(const uint8_t *p, FqC *out) {
    FqC v;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *q = p + 28 - 4 * i;
        v.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    bool below = false;
    for (int i = 7; i >= 0; i--) {
        if (v.v[i] < FqParams::p(i)) { below = true; break; }
        if (v.v[i] > FqParams::p(i)) break;
    }
    *out = FqC::to_mont(v);
    return below;
}
static __device__ __forceinline__ bool rest_is_zero(const uint8_t *p, int n) {      // infinity encoding: 0x40, then zeros
    uint32_t o = p[0] & 0x3f;
    for (int i = 1; i < n; i++) o |= p[i];
    return o == 0;
}
static __device__ __forceinline__ FqC fq_const(const uint32_t (&w)[8]) { FqC t; for (int i = 0; i < 8; i++) t.v[i] = w[i]; return t; }

// in: n points of 64 bytes (BE x || y); out: raw Montgomery LE, infinity -> zeros.  Always: flag bits, coordinates < q, a clean
// infinity encoding (what pairing_ce's `into_affine_unchecked` enforces).  FK_KEY_CHECKED: y^2 = x^3 + 3 (the cofactor of G1 is
// 1, so on the curve is in the group).  FK_KEY_NO_INFINITY: no identity points.
__global__ void convert_g1_kernel(const uint8_t *in, size_t n, Affine<FqC> *out, uint32_t flags, KeyErr *err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in + 64 * i;
    if (p[0] & 0x40) {
        if (!rest_is_zero(p, 64)) atomicAdd(&err->inf_rest, 1u);
        if (flags & FK_KEY_NO_INFINITY) atomicAdd(&err->infinity, 1u);
        out[i] = Affine<FqC>::inf();
        return;
    }
    if (p[0] & 0x80) atomicAdd(&err->flag, 1u);          // compression flag in an uncompressed encoding
    Affine<FqC> a;
    const bool okx = be32_to_mont(p, &a.x), oky = be32_to_mont(p + 32, &a.y), ok = okx && oky;
    if (!ok) atomicAdd(&err->range, 1u);
    if (ok && (flags & FK_KEY_CHECKED)) {
        const uint32_t bw[8] = FK_G1_B;
        const FqC rhs = FqC::add(FqC::mul(FqC::sqr(a.x), a.x), fq_const(bw));
        if (FqC::sqr(a.y) != rhs) atomicAdd(&err->curve, 1u);
    }
    out[i] = a;
}
// G2: x.c1 || x.c0 || y.c1 || y.c0; curve y^2 = x^3 + 3/(9+u) over Fq2 and -- checked -- membership of the order-r subgroup
// (the twist has a large cofactor): r * P = identity, one double-and-add per point.
__global__ __launch_bounds__(128) void convert_g2_kernel(const uint8_t *in, size_t n, Affine<Fq2C> *out, uint32_t flags, KeyErr *err) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = in + 128 * i;
    if (p[0] & 0x40) {
        if (!rest_is_zero(p, 128)) atomicAdd(&err->inf_rest, 1u);
        if (flags & FK_KEY_NO_INFINITY) atomicAdd(&err->infinity, 1u);
        out[i] = Affine<Fq2C>::inf();
        return;
    }
    if (p[0] & 0x80) atomicAdd(&err->flag, 1u);
    Affine<Fq2C> a;
    const bool ok0 = be32_to_mont(p, &a.x.c1), ok1 = be32_to_mont(p + 32, &a.x.c0), ok2 = be32_to_mont(p + 64, &a.y.c1), ok3 = be32_to_mont(p + 96, &a.y.c0);
    const bool ok = ok0 && ok1 && ok2 && ok3;
    if (!ok) atomicAdd(&err->range, 1u);
    out[i] = a;
    if (ok && (flags & FK_KEY_CHECKED)) {
        const uint32_t b0[8] = FK_G2_B0, b1[8] = FK_G2_B1;
        const Fq2C rhs = Fq2C::add(Fq2C::mul(Fq2C::sqr(a.x), a.x), Fq2C{fq_const(b0), fq_const(b1)});
        if (Fq2C::sqr(a.y) != rhs) { atomicAdd(&err->curve, 1u); return; }
        const uint32_t r[8] = FK_R_CANON;
        if (!Xyzz<Fq2C>::mul_scalar(Xyzz<Fq2C>::from_affine(a), r).is_inf()) atomicAdd(&err->subgroup, 1u);
    }
}

// ---- the other direction: bellman `Parameters::write` (mod.rs:156) from the device layout.  Montgomery LE limbs -> canonical 32-byte
// BIG-endian integers (one Montgomery multiplication by 1, byte reversal); the identity (all zeros here, group.rs:55) -> 0x40 then zeros.
static __device__ __forceinline__ void mont_to_be32(const FqC &m, uint8_t *p) {
    const FqC v = FqC::from_mont(m);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t *q = p + 28 - 4 * i;
        q[0] = (uint8_t)(v.v[i] >> 24); q[1] = (uint8_t)(v.v[i] >> 16); q[2] = (uint8_t)(v.v[i] >> 8); q[3] = (uint8_t)v.v[i];
    }
}
__global__ void write_g1_kernel(const Affine<FqC> *in, size_t n, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t *p = out + 64 * i;
    const Affine<FqC> a = in[i];
    if (a.x.is_zero() && a.y.is_zero()) { for (int k = 0; k < 64; k++) p[k] = 0; p[0] = 0x40; return; }
    mont_to_be32(a.x, p); mont_to_be32(a.y, p + 32);
}
__global__ void write_g2_kernel(const Affine<Fq2C> *in, size_t n, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t *p = out + 128 * i;
    const Affine<Fq2C> a = in[i];
    if (a.x.is_zero() && a.y.is_zero()) { for (int k = 0; k < 128; k++) p[k] = 0; p[0] = 0x40; return; }
    mont_to_be32(a.x.c1, p); mont_to_be32(a.x.c0, p + 32); mont_to_be32(a.y.c1, p + 64); mont_to_be32(a.y.c0, p + 96);      // c1 before c0
}

struct Cursor {
    const uint8_t *p; size_t left;
    bool take(size_t n, const uint8_t **out) { if (left < n) return false; *out = p; p += n; left -= n; return true; }
    bool u32be(uint32_t *v) { const uint8_t *q; if (!take(4, &q)) return false; *v = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; return true; }
};

}  // namespace fk

using namespace fk;

extern "C" {

// buf/len: bellman `Parameters::write` bytes.  ic_out (may be NULL): receives up to ic_cap raw 64-byte points;
// *n_ic gets the count; gamma_g2_out (may be NULL): 128 bytes raw.  shard arguments as in fk_key_desc.
int fk_key_load_bellman(fk_ctx *ctx, const uint8_t *buf, size_t len, uint32_t flags, uint32_t shard_index, uint32_t shard_count, double z_frac_lo,
                        double z_frac_hi, fk_key **out, uint8_t *gamma_g2_out, uint8_t *ic_out, uint32_t ic_cap, uint32_t *n_ic) { return fk_guard(ctx, [&]() -> int {
    FK_RANGE("fk_key_load_bellman");
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!buf || !out) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key file: null argument");
    if (flags & ~(uint32_t)(FK_KEY_CHECKED | FK_KEY_NO_INFINITY | FK_KEY_NO_LEVELS)) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key file: unknown flags 0x%x", flags);
    *out = nullptr;
    FK_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_start = std::chrono::steady_clock::now();
    Cursor c{buf, len};
    const uint8_t *vkp[6];
    const size_t vkw[6] = {64, 64, 128, 128, 64, 128};    // alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2
    for (int i = 0; i < 6; i++) if (!c.take(vkw[i], &vkp[i])) FK_SET_ERR(ctx, FK_ERR_FORMAT, "key file: truncated verifying key");
    uint32_t cnt[6]; const uint8_t *arr[6];
    const size_t w[6] = {64, 64, 64, 64, 64, 128};        // ic, h, l, a, b_g1, b_g2
    static const char *nm[6] = {"ic", "h", "l", "a", "b_g1", "b_g2"};
    for (int i = 0; i < 6; i++) {
        if (!c.u32be(&cnt[i]) || !c.take((size_t)cnt[i] * w[i], &arr[i])) FK_SET_ERR(ctx, FK_ERR_FORMAT, "key file: truncated %s array", nm[i]);
    }
    if (cnt[0] == 0) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key file: empty ic (no constant ONE input)");
    if (cnt[4] != cnt[5]) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key file: b_g1 and b_g2 differ in length");
    const uint64_t m = (uint64_t)cnt[1] + 1;
    if (m & (m - 1)) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key file: h holds %u points, not 2^k - 1", cnt[1]);
    if (m > ((uint64_t)1 << (FK_FR_S - 1))) FK_SET_ERR(ctx, FK_ERR_DOMAIN_TOO_LARGE, "key file: domain exceeds 2^%d", FK_FR_S - 1);
    if (shard_count == 0 || shard_index >= shard_count) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key file: bad shard %u/%u", shard_index, shard_count);

    fk_key *k = new fk_key();
    k->m = m; k->num_input = cnt[0]; k->num_aux = cnt[2];
    k->n_h = cnt[1]; k->n_l = cnt[2]; k->n_a = cnt[3]; k->n_b = cnt[4];
    k->shard_index = shard_index; k->shard_count = shard_count;
    { const int rcs = key_plan_slices(ctx, k, z_frac_lo, z_frac_hi); if (rcs != FK_OK) { delete k; return rcs; } }
    auto fail = [&](int code, const char *msg) { ctx->err = msg; fk_key_free(ctx, k); return code; };

    KeyErr *d_bad = nullptr;
    if (hipMalloc((void **)&d_bad, sizeof(KeyErr)) != hipSuccess) return fail(FK_ERR_OOM, "key file: device allocation failed");
    (void)hipMemset(d_bad, 0, sizeof(KeyErr));
    // staged conversion: raw big-endian bytes go up in chunks, converted points are written in place
    const size_t CH = (size_t)1 << 22;   // points per chunk
    auto conv = [&](const uint8_t *src, uint64_t lo, uint64_t hi, size_t width, void **dst) -> int {
        const uint64_t n = hi - lo;
        if (hipMalloc(dst, (n + 1) * width) != hipSuccess) return FK_ERR_OOM;
        for (uint64_t off = 0; off < n; off += CH) {
            const size_t cn = (size_t)((n - off) < CH ? (n - off) : CH);
            if (ctx->misc.reserve(cn * width) != hipSuccess) return FK_ERR_OOM;
            if (hipMemcpyAsync(ctx->misc.p, src + (lo + off) * width, cn * width, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return FK_ERR_HIP;
            if (width == 64) hipLaunchKernelGGL(convert_g1_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, ctx->stream, ctx->misc.as<uint8_t>(), cn, (Affine<FqC> *)*dst + off, flags & (FK_KEY_CHECKED | FK_KEY_NO_INFINITY), d_bad);
            else hipLaunchKernelGGL(convert_g2_kernel, dim3((unsigned)((cn + 127) / 128)), dim3(128), 0, ctx->stream, ctx->misc.as<uint8_t>(), cn, (Affine<Fq2C> *)*dst + off, flags & (FK_KEY_CHECKED | FK_KEY_NO_INFINITY), d_bad);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return FK_ERR_HIP;   // misc is reused by the next chunk
        }
        return FK_OK;
    };
    int rc = conv(arr[1], k->h_lo, k->h_hi, 64, (void **)&k->d_h);
    if (rc == FK_OK) rc = conv(arr[2], k->l_lo, k->l_hi, 64, (void **)&k->d_l);
    if (rc == FK_OK) rc = conv(arr[3], k->a_lo, k->a_hi, 64, (void **)&k->d_a);
    if (rc == FK_OK) rc = conv(arr[4], k->b_lo, k->b_hi, 64, (void **)&k->d_b1);
    if (rc == FK_OK) rc = conv(arr[5], k->b2_lo, k->b2_hi, 128, (void **)&k->d_b2);
    // vk + ic through the same kernels
    G1Affine vk1[3]; G2Affine vk2[3];
    std::vector<G1Affine> ic(cnt[0]);
    if (rc == FK_OK) {
        void *d_tmp = nullptr;
        std::vector<uint8_t> g1buf(3 * 64 + (size_t)cnt[0] * 64), g2buf(3 * 128);
        memcpy(g1buf.data(), vkp[0], 64); memcpy(g1buf.data() + 64, vkp[1], 64); memcpy(g1buf.data() + 128, vkp[4], 64);
        memcpy(g1buf.data() + 192, arr[0], (size_t)cnt[0] * 64);
        memcpy(g2buf.data(), vkp[2], 128); memcpy(g2buf.data() + 128, vkp[3], 128); memcpy(g2buf.data() + 256, vkp[5], 128);
        const size_t n1 = 3 + cnt[0];
        if (hipMalloc(&d_tmp, g1buf.size() + n1 * 64 + g2buf.size() + 3 * 128) != hipSuccess) rc = FK_ERR_OOM;
        else {
            uint8_t *d_in1 = (uint8_t *)d_tmp; G1Affine *d_o1 = (G1Affine *)(d_in1 + g1buf.size());
            uint8_t *d_in2 = (uint8_t *)(d_o1 + n1); G2Affine *d_o2 = (G2Affine *)(d_in2 + g2buf.size());
            std::vector<G1Affine> o1(n1);
            if (hipMemcpy(d_in1, g1buf.data(), g1buf.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_in2, g2buf.data(), g2buf.size(), hipMemcpyHostToDevice) != hipSuccess) rc = FK_ERR_HIP;
            else {
                // bellman's VerifyingKey::read (restated from upstream, not verifiable here) decodes alpha, beta, gamma, delta and ic
                // with the CHECKED `into_affine` whatever Parameters::read's `checked` says, and refuses the identity among the ic
                // points only
                hipLaunchKernelGGL(convert_g1_kernel, dim3(1), dim3(256), 0, ctx->stream, d_in1, (size_t)3, (Affine<FqC> *)d_o1, (uint32_t)FK_KEY_CHECKED, d_bad);
                hipLaunchKernelGGL(convert_g1_kernel, dim3((unsigned)((cnt[0] + 255) / 256)), dim3(256), 0, ctx->stream, d_in1 + 3 * 64, (size_t)cnt[0], (Affine<FqC> *)(d_o1 + 3),
                                   (uint32_t)(FK_KEY_CHECKED | FK_KEY_NO_INFINITY), d_bad);
                hipLaunchKernelGGL(convert_g2_kernel, dim3(1), dim3(128), 0, ctx->stream, d_in2, (size_t)3, (Affine<Fq2C> *)d_o2, (uint32_t)FK_KEY_CHECKED, d_bad);
                if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(o1.data(), d_o1, n1 * 64, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(vk2, d_o2, 3 * 128, hipMemcpyDeviceToHost) != hipSuccess) rc = FK_ERR_HIP;
                else { vk1[0] = o1[0]; vk1[1] = o1[1]; vk1[2] = o1[2]; for (uint32_t i = 0; i < cnt[0]; i++) ic[i] = o1[3 + i]; }
            }
            (void)hipFree(d_tmp);
        }
    }
    KeyErr bad{};
    if (rc == FK_OK && hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) rc = FK_ERR_HIP;
    (void)hipFree(d_bad);
    if (rc != FK_OK) return fail(rc, "key file: conversion failed");
    if (bad.flag || bad.range || bad.inf_rest || bad.curve || bad.subgroup || bad.infinity) {       // GroupDecodingError
        char msg[320];
        snprintf(msg, sizeof msg, "key file: %u points with the compression flag set, %u coordinates not below q, %u malformed infinity encodings, "
                 "%u points not on the curve, %u G2 points outside the order-r subgroup, %u identity points where none are allowed",
                 bad.flag, bad.range, bad.inf_rest, bad.curve, bad.subgroup, bad.infinity);
        return fail(FK_ERR_FORMAT, msg);
    }
    k->alpha_g1 = vk1[0]; k->beta_g1 = vk1[1]; k->delta_g1 = vk1[2];
    k->beta_g2 = vk2[0]; k->delta_g2 = vk2[2];
    if (gamma_g2_out) memcpy(gamma_g2_out, &vk2[1], 128);
    if (n_ic) *n_ic = cnt[0];
    if (ic_out) memcpy(ic_out, ic.data(), (size_t)(cnt[0] < ic_cap ? cnt[0] : ic_cap) * 64);
    const auto t_arrays = std::chrono::steady_clock::now();
    if (!(flags & FK_KEY_NO_LEVELS) && (rc = key_precompute(ctx, k)) != FK_OK) { fk_key_free(ctx, k); return rc; }
    k->load_s[0] = std::chrono::duration<double>(t_arrays - t_start).count();
    k->load_s[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_arrays).count();
    *out = k;
    return FK_OK;
}); }

// vk points of a key as raw Montgomery LE: alpha_g1, beta_g1, delta_g1 (64 B each) then beta_g2, delta_g2 (128 B each)
int fk_key_vk(const fk_key *key, uint8_t out[3 * 64 + 2 * 128]) {
    if (!key || !out) return FK_ERR_BAD_ARG;
    memcpy(out, &key->alpha_g1, 64); memcpy(out + 64, &key->beta_g1, 64); memcpy(out + 128, &key->delta_g1, 64);
    memcpy(out + 192, &key->beta_g2, 128); memcpy(out + 320, &key->delta_g2, 128);
    return FK_OK;
}

// m, num_input, num_aux, n_h, n_l, n_a, n_b, shard_count
int fk_key_counts(const fk_key *key, uint64_t out[8]) {
    if (!key || !out) return FK_ERR_BAD_ARG;
    const uint64_t v[8] = {key->m, key->num_input, key->num_aux, key->n_h, key->n_l, key->n_a, key->n_b, key->shard_count};
    memcpy(out, v, sizeof v);
    return FK_OK;
}

// bellman `Parameters::write` of a WHOLE key resident in HBM (the counterpart of fk_key_load_bellman; fawkes' own header -- gate count,
// gate blob, const-tracker bits, mod.rs:150-155 -- is the host's: params_io.write_parameters).  A proving key does not hold gamma_g2 and
// the ic points: the caller passes what fk_setup* / fk_key_load_bellman returned.  out == NULL: only *needed is set.
int fk_key_write_bellman(fk_ctx *ctx, const fk_key *key, const uint8_t *gamma_g2, const uint8_t *ic, uint32_t n_ic, uint8_t *out, size_t cap,
                         size_t *needed) { return fk_guard(ctx, [&]() -> int {
    if (!ctx) return FK_ERR_BAD_ARG;
    if (!key || !gamma_g2 || (!ic && n_ic) || !needed) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key write: null argument");
    if (key->shard_count != 1) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key write: a shard cannot be written as a Parameters file (load the whole key)");
    if (n_ic != key->num_input) FK_SET_ERR(ctx, FK_ERR_KEY_MISMATCH, "key write: %u ic points for %u inputs", n_ic, key->num_input);
    const uint64_t cnt[5] = {key->h_hi - key->h_lo, key->l_hi - key->l_lo, key->a_hi - key->a_lo, key->b_hi - key->b_lo, key->b2_hi - key->b2_lo};
    const size_t total = 3 * 64 + 3 * 128 + 4 + (size_t)n_ic * 64 + 5 * 4 + (size_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3]) * 64 + (size_t)cnt[4] * 128;
    *needed = total;
    if (!out) return FK_OK;
    if (cap < total) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key write: buffer of %zu bytes, %zu needed", cap, total);
    FK_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *w = out;
    auto u32be = [&](uint32_t v) { w[0] = (uint8_t)(v >> 24); w[1] = (uint8_t)(v >> 16); w[2] = (uint8_t)(v >> 8); w[3] = (uint8_t)v; w += 4; };
    // device conversion in chunks through ctx->misc: raw points in (device-resident arrays as they are; host points uploaded first)
    // (sized by the key: a key of a few thousand points must not grow the context's scratch by a GiB on a device that is full -- ADVICE r4)
    for (int i = 0; i < 5; i++) if (cnt[i] > 0xffffffffull) FK_SET_ERR(ctx, FK_ERR_BAD_ARG, "key write: array too long for bellman's u32 count");
    const size_t longest = (size_t)std::max<uint64_t>(std::max(std::max(cnt[0], cnt[1]), std::max(cnt[2], std::max(cnt[3], cnt[4]))), std::max<uint64_t>(n_ic, 1));
    const size_t CH = std::min((size_t)1 << 22, longest);          // points per chunk
    FK_HIP(ctx, ctx->misc.reserve(CH * 128 * 2));
    uint8_t *d_bytes = ctx->misc.as<uint8_t>(), *d_stage = d_bytes + CH * 128;
    auto emit = [&](const void *src, bool src_on_device, size_t n, size_t width) -> int {
        for (size_t off = 0; off < n; off += CH) {
            const size_t cn = std::min(CH, n - off);
            const uint8_t *d_src = (const uint8_t *)src + off * width;
            if (!src_on_device) { FK_HIP(ctx, hipMemcpyAsync(d_stage, d_src, cn * width, hipMemcpyHostToDevice, ctx->stream)); d_src = d_stage; }
            if (width == 64) hipLaunchKernelGGL(write_g1_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, ctx->stream, (const Affine<FqC> *)d_src, cn, d_bytes);
            else hipLaunchKernelGGL(write_g2_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, ctx->stream, (const Affine<Fq2C> *)d_src, cn, d_bytes);
            FK_HIP(ctx, hipGetLastError());
            FK_HIP(ctx, hipMemcpyAsync(w, d_bytes, cn * width, hipMemcpyDeviceToHost, ctx->stream));
            FK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            w += cn * width;
        }
        return FK_OK;
    };
    // vk: alpha_g1, beta_g1 (G1), beta_g2, gamma_g2 (G2), delta_g1 (G1), delta_g2 (G2), u32 BE count, ic[]
    FK_TRY(emit(&key->alpha_g1, false, 1, 64)); FK_TRY(emit(&key->beta_g1, false, 1, 64));
    FK_TRY(emit(&key->beta_g2, false, 1, 128)); FK_TRY(emit(gamma_g2, false, 1, 128));
    FK_TRY(emit(&key->delta_g1, false, 1, 64)); FK_TRY(emit(&key->delta_g2, false, 1, 128));
    u32be(n_ic); FK_TRY(emit(ic, false, n_ic, 64));
    // h, l, a, b_g1 (G1), b_g2 (G2): u32 BE count + points
    u32be((uint32_t)cnt[0]); FK_TRY(emit(key->d_h, true, cnt[0], 64));
    u32be((uint32_t)cnt[1]); FK_TRY(emit(key->d_l, true, cnt[1], 64));
    u32be((uint32_t)cnt[2]); FK_TRY(emit(key->d_a, true, cnt[2], 64));
    u32be((uint32_t)cnt[3]); FK_TRY(emit(key->d_b1, true, cnt[3], 64));
    u32be((uint32_t)cnt[4]); FK_TRY(emit(key->d_b2, true, cnt[4], 128));
    if ((size_t)(w - out) != total) FK_SET_ERR(ctx, FK_ERR_HIP, "key write: wrote %zu of %zu bytes", (size_t)(w - out), total);
    return FK_OK;
}); }

}  // extern "C"
